// Latency probes for the serial chains of the solver (one wave): dependent FP64 FMA, v_readlane broadcast + FMA,
// LDS write -> fence -> read round trip.  hipcc --offload-arch=gfx950 -O3 chain.hip -o chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R8(x) x x x x x x x x
__device__ __forceinline__ double lane_f64(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_ror1(double a) {
    int lo = __double2loint(a), hi = __double2hiint(a);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x121, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x121, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__global__ void k_dep(double* out, long long* cyc, int n) {
    double a = out[threadIdx.x], b = 1.0000001, c = 0.5;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) { R8(a = fma(a, b, c);) }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / (8 * n);
}
__global__ void k_bcast(double* out, long long* cyc, int n) {   // a <- m * a[lane 3] + c : readlane pair + FMA per link
    double a = out[threadIdx.x], m = 0.999 + 1e-3 * threadIdx.x, c = 0.5;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) { R8(a = fma(m, lane_f64(a, 3), c);) }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / (8 * n);
}
__global__ void k_bcast6(double* out, long long* cyc, int n) {  // one adjoint stage: 6 broadcasts of the old value, 6-FMA chain
    double a = out[threadIdx.x], m = 0.1 + 1e-3 * threadIdx.x, c = 0.5;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        double t = c;
#pragma unroll
        for (int j = 0; j < 6; j++) t = fma(m, lane_f64(a, j), t);
        a = t;
    }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / n;
}
__global__ void k_lds6(double* out, long long* cyc, int n) {    // the same stage through LDS: write, fence, 6 broadcast reads, 6-FMA chain
    __shared__ double sh[64];
    double a = out[threadIdx.x], m = 0.1 + 1e-3 * threadIdx.x, c = 0.5;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        sh[threadIdx.x] = a;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double t = c;
#pragma unroll
        for (int j = 0; j < 6; j++) t = fma(m, sh[j], t);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        a = t;
    }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / n;
}
__global__ void k_ldsrt(double* out, long long* cyc, int n) {   // bare LDS round trip: write own slot, read neighbour's
    __shared__ double sh[64];
    double a = out[threadIdx.x];
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        R8(sh[threadIdx.x] = a; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); a = sh[(threadIdx.x + 1) & 63] + 1.0; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");)
    }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / (8 * n);
}
__global__ void k_dpp(double* out, long long* cyc, int n) {     // DPP row rotate + FMA link
    double a = out[threadIdx.x], m = 0.999, c = 0.5;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        R8(a = fma(m, dpp_ror1(a), c);)
    }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / (8 * n);
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8); hipMemset(out, 0, 64 * 8);
    long long h;
#define RUN(k, name) k<<<1, 64>>>(out, cyc, 2000); hipDeviceSynchronize(); k<<<1, 64>>>(out, cyc, 2000); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-44s %lld cycles\n", name, h);
    RUN(k_dep, "dependent v_fma_f64, per link");
    RUN(k_bcast, "readlane pair + fma, per link");
    RUN(k_dpp, "dpp pair + fma, per link");
    RUN(k_bcast6, "6 broadcasts + 6-fma chain (register stage)");
    RUN(k_lds6, "write+fence+6 LDS reads+6-fma chain (LDS stage)");
    RUN(k_ldsrt, "LDS write -> read neighbour, per round trip");
    return 0;
}
