// halfrow_dot6 (crx_wave.h): does lane 8 g + c receive sum_j m[j] x(lane 8 g + j)?  (First form, one accumulator + bank_mask 0x3 / 0xc: the lanes of a
// masked-off bank came back ZERO, not unchanged -- measured on gfx950; the shipped form keeps two accumulators and selects.)
// hipcc --offload-arch=gfx950 -O3 -I../../car-racing_amd/csrc halfrow.hip -o halfrow
#include "crx_wave.h"
#include <stdio.h>
__global__ void k(double* out, double* out2, long long* cyc) {
    const int lane = threadIdx.x;
    double x = (double)lane;
    double m[6] = {1.0, 100.0, 1e4, 1e6, 1e8, 1e10};
    asm volatile("" : "+v"(x));
    long long t0 = clock64();
    double a = halfrow_dot6(x, m, (lane & 8) != 0);
    long long t1 = clock64();
    out[lane] = a;
    double mz[7] = {1.0, 100.0, 1e4, 1e6, 1e8, 1e10, 1e12};
    out2[lane] = row_dot<7, 8>(x, mz, 0.5);
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    double *o, *o2; long long* c; (void)hipMalloc(&o, 512); (void)hipMalloc(&o2, 512); (void)hipMalloc(&c, 8);
    k<<<1, 64>>>(o, o2, c); (void)hipDeviceSynchronize();
    double h[64], h2[64]; long long hc; (void)hipMemcpy(h, o, 512, hipMemcpyDeviceToHost); (void)hipMemcpy(h2, o2, 512, hipMemcpyDeviceToHost); (void)hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        double e = 0; const double m[6] = {1.0, 100.0, 1e4, 1e6, 1e8, 1e10};
        for (int j = 0; j < 6; j++) e += m[j] * (double)((l & ~7) + j);
        if (e != h[l]) { bad++; printf("lane %2d: got %.1f expected %.1f\n", l, h[l], e); }
    }
    for (int l = 0; l < 64; l++) {
        double e = 0.5; const double m[7] = {1.0, 100.0, 1e4, 1e6, 1e8, 1e10, 1e12};
        for (int j = 0; j < 7; j++) e += m[j] * (double)((l & ~15) + 8 + j);
        if (e != h2[l]) { bad++; printf("row_dot<7,8> lane %2d: got %.1f expected %.1f\n", l, h2[l], e); }
    }
    printf("halfrow_dot6: %d mismatches, %lld ticks\n", bad, hc);
    return bad != 0;
}
