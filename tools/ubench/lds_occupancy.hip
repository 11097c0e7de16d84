// Resident single-wave workgroups per CU as a function of the dynamic LDS size (the runtime's answer): finds the LDS
// allocation granularity that decides how many problems of a given footprint share a CU.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_occupancy.hip -o /tmp/lds_occ && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) probe(double* out) {
    extern __shared__ double sm[];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (out) out[threadIdx.x] = sm[63 - threadIdx.x];
}
int main() {
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int last = -1;
    for (int bytes = 8192; bytes <= 20480; bytes += 64) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, probe, 64, bytes) != hipSuccess) { printf("error at %d\n", bytes); return 1; }
        if (n != last) printf("%6d B -> %d per CU\n", bytes, n);
        last = n;
    }
    return 0;
}
