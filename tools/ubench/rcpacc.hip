// accuracy of v_rcp_f64 raw / after one / two Newton steps, and latency of the three variants as a dependent chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k_acc(double* out) {
    double worst[3] = {0, 0, 0};
    unsigned long long st = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1);
    for (int i = 0; i < 200000; i++) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        double x = __longlong_as_double((long long)((st & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull)) * (1.0 + (st >> 60));
        double r0 = __builtin_amdgcn_rcp(x);
        double r1 = fma(r0, fma(-x, r0, 1.0), r0);
        double r2 = fma(r1, fma(-x, r1, 1.0), r1);
        double ex = 1.0 / x;
        worst[0] = fmax(worst[0], fabs(r0 - ex) / ex); worst[1] = fmax(worst[1], fabs(r1 - ex) / ex); worst[2] = fmax(worst[2], fabs(r2 - ex) / ex);
    }
    for (int q = 0; q < 3; q++) out[threadIdx.x * 3 + q] = worst[q];
}
template <int NS>
__global__ void k_lat(double* out, long long* cyc, int n) {
    double a = out[threadIdx.x] + 2.0;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            double r = __builtin_amdgcn_rcp(a);
            if (NS >= 1) r = fma(r, fma(-a, r, 1.0), r);
            if (NS >= 2) r = fma(r, fma(-a, r, 1.0), r);
            a = r + 1.5;
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = (t1 - t0) / (8 * n);
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 64 * 3 * 8); hipMalloc(&cyc, 8); hipMemset(out, 0, 64 * 3 * 8);
    k_acc<<<1, 64>>>(out); double h[192]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    double w[3] = {0, 0, 0}; for (int i = 0; i < 64; i++) for (int q = 0; q < 3; q++) w[q] = fmax(w[q], h[i * 3 + q]);
    printf("max rel err: raw %.3e, 1 Newton %.3e, 2 Newton %.3e\n", w[0], w[1], w[2]);
    long long c;
    hipMemset(out, 0, 64 * 8); k_lat<0><<<1, 64>>>(out, cyc, 1000); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("rcp + add chain link: raw %lld", c);
    hipMemset(out, 0, 64 * 8); k_lat<1><<<1, 64>>>(out, cyc, 1000); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf(", 1 Newton %lld", c);
    hipMemset(out, 0, 64 * 8); k_lat<2><<<1, 64>>>(out, cyc, 1000); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf(", 2 Newton %lld cycles\n", c);
    return 0;
}
