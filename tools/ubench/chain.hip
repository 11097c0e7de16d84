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
__global__ void k_indep(double* out, long long* cyc, int n) {   // independent FP64 FMAs (8 accumulators): the issue rate of one wave
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0000001, c = 0.5;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) { a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c); }
    long long t1 = clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / (8 * n);
}
__global__ void k_indep_i32(double* out, long long* cyc, int n) {   // independent 32-bit integer adds: the issue rate of plain VALU
    int a0 = (int)out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = threadIdx.x;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) { a0 = a0 * 3 + b; a1 = a1 * 3 + b; a2 = a2 * 3 + b; a3 = a3 * 3 + b; a4 = a4 * 3 + b; a5 = a5 * 3 + b; a6 = a6 * 3 + b; a7 = a7 * 3 + b; asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    long long t1 = clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / (8 * n);
}
__global__ void k_dppfmac(double* out, long long* cyc, int n) {   // v_fmac_f64_dpp row_newbcast chain (7 links + s_nop 1), per link
    double a = out[threadIdx.x], m = 0.1 + 1e-3 * threadIdx.x, c = 0.5;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        double t = c;
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(a), "v"(m));
        a = t;
    }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / n;
}
__global__ void k_rcp(double* out, long long* cyc, int n) {   // the pivot chain: v_rcp_f64 + two Newton steps, dependent
    double a = out[threadIdx.x] + 1.5;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        R8({ double r = __builtin_amdgcn_rcp(a); double e = fma(-a, r, 1.0); r = fma(r, e, r); e = fma(-a, r, 1.0); r = fma(r, e, r); a = r + 1.25; })
    }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / (8 * n);
}
__global__ void k_salu(double* out, long long* cyc, int n) {   // scalar adds between VALU ops: does a scalar instruction cost the wave an issue slot?
    double a = out[threadIdx.x], b = 1.0000001, c = 0.5; int s = n;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) { R8(a = fma(a, b, c); asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));) }
    long long t1 = clock64();
    out[threadIdx.x] = a + s; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / (8 * n);
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8); hipMemset(out, 0, 64 * 8);
    long long h;
#define RUN(k, name) k<<<1, 64>>>(out, cyc, 2000); hipDeviceSynchronize(); k<<<1, 64>>>(out, cyc, 2000); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-44s %lld cycles\n", name, h);
    RUN(k_dep, "dependent v_fma_f64, per link");
    RUN(k_indep, "independent v_fma_f64 (8 chains), per instr");
    RUN(k_indep_i32, "independent v_mad_u32 (8 chains), per instr");
    RUN(k_salu, "dependent v_fma_f64 + one s_add each, per pair");
    RUN(k_dppfmac, "s_nop 1 + 7 v_fmac_f64_dpp row_newbcast (stage)");
    RUN(k_rcp, "v_rcp_f64 + 2 Newton steps + add, per pivot");
    RUN(k_bcast, "readlane pair + fma, per link");
    RUN(k_dpp, "dpp pair + fma, per link");
    RUN(k_bcast6, "6 broadcasts + 6-fma chain (register stage)");
    RUN(k_lds6, "write+fence+6 LDS reads+6-fma chain (LDS stage)");
    RUN(k_ldsrt, "LDS write -> read neighbour, per round trip");
    return 0;
}
