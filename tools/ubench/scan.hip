// Wave64 exclusive prefix / suffix sums with DPP (row_shr / row_shl + wave_shr:1 / wave_shl:1) -- correctness and latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
template <int CTRL, bool BC>
__device__ __forceinline__ double dppz(double v) {   // lanes without a source get 0 (old = 0, bound_ctrl as given)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, BC);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, BC);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rl(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double incl_prefix(double v, int lane) {
    v += dppz<0x111, true>(v);  // row_shr:1
    v += dppz<0x112, true>(v);  // row_shr:2
    v += dppz<0x114, true>(v);  // row_shr:4
    v += dppz<0x118, true>(v);  // row_shr:8
    const double r0 = rl(v, 15), r1 = rl(v, 31), r2 = rl(v, 47);
    const int row = lane >> 4;
    return v + (row >= 1 ? r0 : 0.0) + (row >= 2 ? r1 : 0.0) + (row >= 3 ? r2 : 0.0);
}
__device__ __forceinline__ double excl_prefix(double v, int lane) { return incl_prefix(dppz<0x138, true>(v), lane); }   // wave_shr:1
__device__ __forceinline__ double incl_suffix(double v, int lane) {
    v += dppz<0x101, true>(v);  // row_shl:1
    v += dppz<0x102, true>(v);
    v += dppz<0x104, true>(v);
    v += dppz<0x108, true>(v);
    const double r1 = rl(v, 16), r2 = rl(v, 32), r3 = rl(v, 48);
    const int row = lane >> 4;
    return v + (row <= 2 ? r3 : 0.0) + (row <= 1 ? r2 : 0.0) + (row <= 0 ? r1 : 0.0);
}
__device__ __forceinline__ double excl_suffix(double v, int lane) { return incl_suffix(dppz<0x130, true>(v), lane); }   // wave_shl:1
__global__ void k(double* out, long long* cyc) {
    const int lane = threadIdx.x;
    const double x = 1.0 + lane * 0.5;
    out[lane] = excl_prefix(x, lane);
    out[64 + lane] = excl_suffix(x, lane);
    out[128 + lane] = incl_prefix(x, lane);
    double a = x;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; i++) a = excl_prefix(a * 1e-3, lane) + 1.0;
    long long t1 = clock64();
    out[192 + lane] = a;
    if (lane == 0) cyc[0] = (t1 - t0) / 1000;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 8);
    k<<<1, 64>>>(out, cyc); double h[256]; long long c; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        double ep = 0, es = 0;
        for (int j = 0; j < i; j++) ep += 1.0 + j * 0.5;
        for (int j = i + 1; j < 64; j++) es += 1.0 + j * 0.5;
        if (fabs(h[i] - ep) > 1e-9 || fabs(h[64 + i] - es) > 1e-9 || fabs(h[128 + i] - (ep + 1.0 + i * 0.5)) > 1e-9) { bad++; if (bad < 5) printf("lane %d: excl %g (want %g) suffix %g (want %g)\n", i, h[i], ep, h[64 + i], es); }
    }
    printf("scan check: %d bad lanes; dependent excl_prefix + mul + add link: %lld cycles\n", bad, c);
    return 0;
}
