// Round 5, VERDICT item 2 ("more than one wave per problem for small launches"): what does a PHASE BOUNDARY between the waves of one
// workgroup cost?  One wave per problem needs none (lockstep lanes + in-order LDS: SYNC() is a fence, no instruction); W waves per problem
// need, at every one of the ~40 SYNC() points of an interior-point iteration, the release/acquire pair
//     ds_write ... ; s_waitcnt lgkmcnt(0) ; s_barrier ; ds_read ... ; s_waitcnt lgkmcnt(0)
// Measured here for W = 1, 2, 4 waves (one per SIMD of a CU) in s_memtime ticks AND nanoseconds (hipEvent around the kernel), next to the
// same exchange inside ONE wave, a bare s_barrier, and -- for VERDICT item 4's A/B -- a dependent chain of v_mfma_f64_16x16x4_f64.
//     hipcc --offload-arch=gfx950 -O3 barrier.hip -o barrier && ./barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));

// exchange through LDS with a real barrier: every thread writes its slot, reads the slot of the thread 64 further (the next wave)
template <int W, bool BARRIER>
__global__ void __launch_bounds__(64 * W) k_exchange(double* out, long long* cyc, int n) {
    __shared__ double sm[64 * W];
    const int t = threadIdx.x, peer = (t + 64) % (64 * W);
    double v = out[t];
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        sm[t] = v;
        if (BARRIER) __syncthreads();                       // s_waitcnt lgkmcnt(0) + s_barrier
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        v = sm[peer] * 1.0000001 + 0.5;                     // the data dependence keeps one exchange per trip
        if (BARRIER) __syncthreads();                       // (a second boundary: the slot is rewritten next trip)
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    const long long t1 = clock64();
    out[t] = v;
    if (t == 0) cyc[0] = (t1 - t0) * 10 / n;
}
template <int W>
__global__ void __launch_bounds__(64 * W) k_barrier_only(double* out, long long* cyc, int n) {
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) { asm volatile("s_barrier"); asm volatile("s_barrier"); asm volatile("s_barrier"); asm volatile("s_barrier"); }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { cyc[0] = (t1 - t0) * 10 / (4 * n); out[0] += 1.0; }
}
// dependent chain of FP64 MFMAs (D feeds C of the next): the latency a lone wave pays per instruction of the T = P M / H = M'T products
__global__ void __launch_bounds__(64) k_mfma_dep(double* out, long long* cyc, int n) {
    const double a = out[threadIdx.x], b = a + 1.0;
    double4_t c = {0.0, 0.0, 0.0, 0.0};
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[threadIdx.x] = c[0] + c[1] + c[2] + c[3];
    if (threadIdx.x == 0) cyc[0] = (t1 - t0) * 10 / (4 * n);
}
// ... and independent ones (four accumulators): the issue cost
__global__ void __launch_bounds__(64) k_mfma_ind(double* out, long long* cyc, int n) {
    const double a = out[threadIdx.x], b = a + 1.0;
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) cyc[0] = (t1 - t0) * 10 / (4 * n);
}
// reference point: a dependent v_fma_f64 chain of one wave (tools/ubench/issue.hip: ~6.5 ticks per instruction whatever its class)
__global__ void __launch_bounds__(64) k_fma_dep(double* out, long long* cyc, int n) {
    double a = out[threadIdx.x];
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) { asm volatile("v_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2\n\tv_fmac_f64 %0, %1, %2" : "+v"(a) : "v"(1.0000001), "v"(0.5)); }
    const long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = (t1 - t0) * 10 / (4 * n);
}
int main() {
    double* out; long long* cyc; (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&cyc, 8); (void)hipMemset(out, 0, 256 * 8);
    long long h; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); float ms;
    const int n = 4000;
#define RUN(launch, per, name) launch; (void)hipDeviceSynchronize(); (void)hipEventRecord(e0); launch; (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); \
    (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-78s %7.1f ticks  %7.1f ns\n", name, h / 10.0, ms * 1e6 / (per));
    RUN((k_fma_dep<<<1, 64>>>(out, cyc, n)), 4.0 * n, "v_fmac_f64, dependent chain, one wave (per instruction)");
    RUN((k_exchange<1, false><<<1, 64>>>(out, cyc, n)), 1.0 * n, "LDS exchange inside ONE wave, fence only (per write -> read round trip)");
    RUN((k_exchange<1, true><<<1, 64>>>(out, cyc, n)), 1.0 * n, "the same with __syncthreads() (one wave: the barrier is elided or trivial)");
    RUN((k_exchange<2, true><<<1, 128>>>(out, cyc, n)), 1.0 * n, "LDS exchange between 2 waves: write, wait, s_barrier, read, s_barrier");
    RUN((k_exchange<4, true><<<1, 256>>>(out, cyc, n)), 1.0 * n, "LDS exchange between 4 waves (one per SIMD)");
    RUN((k_barrier_only<1><<<1, 64>>>(out, cyc, n)), 4.0 * n, "bare s_barrier, 1 wave");
    RUN((k_barrier_only<2><<<1, 128>>>(out, cyc, n)), 4.0 * n, "bare s_barrier, 2 waves");
    RUN((k_barrier_only<4><<<1, 256>>>(out, cyc, n)), 4.0 * n, "bare s_barrier, 4 waves");
    RUN((k_mfma_dep<<<1, 64>>>(out, cyc, n)), 4.0 * n, "v_mfma_f64_16x16x4_f64, dependent chain (per instruction)");
    RUN((k_mfma_ind<<<1, 64>>>(out, cyc, n)), 4.0 * n, "v_mfma_f64_16x16x4_f64, four independent accumulators (per instruction)");
    // the same with 256 workgroups in flight (one per CU, as in the headline launch): does the clock / arbitration change the picture?
    RUN((k_exchange<1, false><<<256, 64>>>(out, cyc, n)), 1.0 * n, "x256 workgroups: LDS exchange inside one wave, fence only");
    RUN((k_exchange<4, true><<<256, 256>>>(out, cyc, n)), 1.0 * n, "x256 workgroups: LDS exchange between 4 waves");
    RUN((k_fma_dep<<<256, 64>>>(out, cyc, n)), 4.0 * n, "x256 workgroups: v_fmac_f64 dependent chain");
    return 0;
}
