// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for libcrx's access pattern (MI355X_MICROARCH.md, HBM section:
// "Other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
//
// One single-wave workgroup per record, exactly as crx_solve_kernel: lane l reads doubles l, l+64, ... of its record
// (8 B per lane, consecutive lanes consecutive addresses) and writes its result record the same way.  Known byte counts:
//   read8  <records> <doubles_in>   reads records*doubles_in*8 B, writes records*8 B
//   write8 <records> <doubles_out>  reads nothing, writes records*doubles_out*8 B
// Run each under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) -- tools/gpu_calib.sh -- at a size
// far beyond the 256 MiB Infinity Cache and at the solver's own sizes (256 x 39 in / 118 out doubles = cfg2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void __launch_bounds__(64) read8(const double* in, double* out, int nd) {
    const double* rec = in + (size_t)blockIdx.x * nd;
    double acc = 0.0;
    for (int e = threadIdx.x; e < nd; e += 64) acc += rec[e];
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(64) write8(double* out, int nd) {
    double* rec = out + (size_t)blockIdx.x * nd;
    for (int e = threadIdx.x; e < nd; e += 64) rec[e] = (double)(blockIdx.x + e);
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s read8|write8 <records> <doubles> [reps]\n", argv[0]); return 2; }
    const int rd = !strcmp(argv[1], "read8");
    const int n = atoi(argv[2]), nd = atoi(argv[3]), reps = argc > 4 ? atoi(argv[4]) : 3;
    double *a, *b;
    const size_t bytes = (size_t)n * nd * 8;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, (size_t)n * 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(a, 0, bytes);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < reps; r++) {
        if (rd) hipLaunchKernelGGL(read8, dim3(n), dim3(64), 0, 0, a, b, nd);
        else hipLaunchKernelGGL(write8, dim3(n), dim3(64), 0, 0, a, nd);
        (void)hipDeviceSynchronize();
    }
    printf("%s records=%d doubles=%d bytes_per_launch=%zu reps=%d\n", argv[1], n, nd, bytes, reps);
    (void)hipFree(a); (void)hipFree(b);
    return 0;
}
