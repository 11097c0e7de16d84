// Issue rate of ONE wave, by instruction class (s_memtime clocks per instruction): 16 independent instructions per trip, inline assembly so that
// the compiler can neither fuse nor reorder them.  hipcc --offload-arch=gfx950 -O3 issue.hip -o issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R4(x) x x x x
#define KERNEL(name, BODY)                                                                          \
    __global__ void name(double* out, long long* cyc, int n) {                                      \
        double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 0.5; \
        float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, g = 1.0000001f, h = 0.5f;       \
        int i0 = (int)a0, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, s0 = n;                             \
        long long t0 = clock64();                                                                   \
        _Pragma("unroll 1") for (int i = 0; i < n; i++) { R4(BODY) }                                 \
        long long t1 = clock64();                                                                   \
        out[threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + i0 + i1 + i2 + i3 + s0;          \
        if (threadIdx.x == 0) cyc[0] = (t1 - t0) * 10 / (16 * n);                                   \
    }
KERNEL(k_f64, asm volatile("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_f64dep, asm volatile("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %0, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_f32, asm volatile("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));)
KERNEL(k_i32, asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i3));)
KERNEL(k_cnd, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i3) : "vcc");)
KERNEL(k_mix, asm volatile("v_fmac_f64 %0, %4, %5\n\tv_add_u32 %2, %2, %3\n\tv_fmac_f64 %1, %4, %5\n\tv_add_u32 %3, %3, %2" : "+v"(a0), "+v"(a1), "+v"(i0), "+v"(i1) : "v"(b), "v"(c));)
KERNEL(k_mixs, asm volatile("v_fmac_f64 %0, %3, %4\n\ts_add_u32 %2, %2, 3\n\tv_fmac_f64 %1, %3, %4\n\ts_add_u32 %2, %2, 5" : "+v"(a0), "+v"(a1), "+s"(s0) : "v"(b), "v"(c));)
KERNEL(k_salu, asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5\n\ts_add_u32 %0, %0, 7\n\ts_add_u32 %0, %0, 9" : "+s"(s0));)
KERNEL(k_mul64, asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %5\n\tv_add_f64 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_mov64, asm volatile("v_mov_b64 %0, %4\n\tv_mov_b64 %1, %4\n\tv_mov_b64 %2, %5\n\tv_mov_b64 %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_rdl, asm volatile("v_readlane_b32 %0, %1, 3\n\tv_readlane_b32 %0, %2, 4\n\tv_readlane_b32 %0, %1, 5\n\tv_readlane_b32 %0, %2, 6" : "+s"(s0) : "v"(i0), "v"(i1));)
KERNEL(k_nop, asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");)
KERNEL(k_wait, asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt lgkmcnt(0)");)
int main() {
    double* out; long long* cyc; (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8); (void)hipMemset(out, 0, 64 * 8);
    long long h;
#define RUN(k, name) k<<<1, 64>>>(out, cyc, 4000); (void)hipDeviceSynchronize(); k<<<1, 64>>>(out, cyc, 4000); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-52s %5.1f clocks\n", name, h / 10.0);
    RUN(k_f64, "v_fmac_f64, four independent accumulators");
    RUN(k_f64dep, "v_fmac_f64, one dependent chain");
    RUN(k_mul64, "v_mul_f64 / v_add_f64, independent");
    RUN(k_f32, "v_fmac_f32, independent");
    RUN(k_i32, "v_add_u32, independent");
    RUN(k_cnd, "v_cndmask_b32, independent");
    RUN(k_mov64, "v_mov_b64, independent");
    RUN(k_mix, "v_fmac_f64 alternating with v_add_u32");
    RUN(k_mixs, "v_fmac_f64 alternating with s_add_u32");
    RUN(k_salu, "s_add_u32, dependent");
    RUN(k_rdl, "v_readlane_b32");
    RUN(k_nop, "s_nop 0");
    RUN(k_wait, "s_waitcnt lgkmcnt(0) (nothing outstanding)");
    return 0;
}
