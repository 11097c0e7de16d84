// ATTEMPT at a cut-down reproducer of the first of the two compiler behaviours of DESIGN.md section 5.7 -- it does NOT reproduce.
// The pattern: lanes of ONE wavefront exchange data through LDS; the only thing between the store of phase 1 and the load of phase 2 is a
// WAVEFRONT-scope fence, which lowers to no instruction.  To LLVM the 64 lanes are independent threads: the load address (sm[lane + 12]) provably
// differs from this thread's own store address (sm[lane]), so alias analysis says "no alias" and nothing but the fence keeps the machine scheduler
// from moving the load above the store (where it would return what the slot held BEFORE lane + 12 wrote it).  In libcrx's 2500-line solver kernel,
// built with -amdgpu-sched-strategy=iterative-ilp and alias analysis in codegen, the scheduler did move such a load in two A/B builds of round 4
// (profiles/r05_wrong_result_hunt.txt: reproduced from their source states in round 5; fixed by the wave barrier, SYNC_KIND 1 below).  HERE, with
// ROCm 7.2's hipcc, default or iterative-ilp scheduling, the ds_read stays behind the ds_write in the emitted code for both SYNC kinds (compile with
// --cuda-device-only -S and look at the order): the fence is honoured in a kernel this small.  What made it fail in the big kernel (register
// pressure steering the iterative scheduler, the particular mix of LDS traffic) was not isolated; the file is kept as the statement of the pattern.
//     hipcc -O3 --offload-arch=gfx950 [-mllvm -amdgpu-sched-strategy=iterative-ilp] -DSYNC_KIND=0 fence_reorder.hip -o fence_reorder && ./fence_reorder
//     SYNC_KIND 0: fence only (libcrx <= 0.2.1)   1: fence + llvm.amdgcn.wave.barrier (libcrx since 0.3.0)
// On a GPU the program prints how many lanes read a stale value (0 expected for either kind with this toolchain).
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef SYNC_KIND
#define SYNC_KIND 0
#endif
#if SYNC_KIND == 0
#define SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#else
#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

__global__ void __launch_bounds__(64) exchange(const double* in, double* out, int rounds) {
    __shared__ double sm[128];
    const int lane = threadIdx.x;
    sm[lane] = -1.0; sm[lane + 64] = -1.0;
    SYNC();
    double acc = 0.0;
    for (int r = 0; r < rounds; r++) {
        // phase 1: a dependent chain (something worth hiding a load behind), then every lane publishes its value
        double v = in[lane] + (double)r;
#pragma unroll
        for (int i = 0; i < 24; i++) v = fma(v, 1.0000001, 0.5);
        sm[lane] = v;
        SYNC();
        // phase 2: every lane reads what lane + 12 published (another address than its own store: "no alias" for one thread)
        const double w = sm[(lane + 12) & 63 ? lane + 12 : lane + 12];
        acc += w;
        SYNC();
    }
    out[lane] = acc;
}

int main() {
    double *in, *out, h_in[64], h_out[64];
    for (int i = 0; i < 64; i++) h_in[i] = 1.0 + i;
    hipMalloc(&in, sizeof(h_in)); hipMalloc(&out, sizeof(h_out));
    hipMemcpy(in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    const int rounds = 8;
    hipLaunchKernelGGL(exchange, dim3(1), dim3(64), 0, 0, in, out, rounds);
    hipMemcpy(h_out, out, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        double ref = 0.0;
        for (int r = 0; r < rounds; r++) {
            const int src = l + 12;                       // lanes 52..63 read the never-written upper half: -1
            double v = src < 64 ? h_in[src] + (double)r : -1.0;
            if (src < 64) for (int i = 0; i < 24; i++) v = fma(v, 1.0000001, 0.5);
            ref += v;
        }
        if (h_out[l] != ref) bad++;
    }
    printf("SYNC_KIND %d: %d of 64 lanes read a stale value\n", SYNC_KIND, bad);
    return bad != 0;
}
