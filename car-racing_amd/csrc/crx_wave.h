// crx_wave.h -- wave-level primitives shared by the gfx950 kernels of libcrx (crx_kernels.hip,
// crx_lmpc.hip).  64-lane wavefronts only.
#ifndef CRX_WAVE_H
#define CRX_WAVE_H
#include <hip/hip_runtime.h>
#include <math.h>

#define WAVE 64

// `while (s > L) s -= L` of the reference (overtake_traj_planner.py:216-217, :291-292; racing_env.py curvature lookup)
// with a bounded trip count: the first 4 laps are subtracted one by one (bit-identical to the reference's loop, which
// never sees more than two), anything beyond is reduced in closed form, and a non-finite or non-positive-lap input
// falls straight through -- garbage in a device-resident array must not be able to hang the GPU.
__device__ __forceinline__ double wrap_above(double s, double L) {
#pragma unroll 1
    for (int i = 0; i < 4 && s > L; i++) s -= L;
    if (s > L && L > 0.0 && s < 1e300) { s -= L * (ceil(s / L) - 1.0); if (s > L) s -= L; }
    return s;
}
__device__ __forceinline__ double wrap_below(double s, double L) {   // `while (s < 0) s += L`
#pragma unroll 1
    for (int i = 0; i < 4 && s < 0.0; i++) s += L;
    if (s < 0.0 && L > 0.0 && s > -1e300) { s += L * ceil(-s / L); }
    return s;
}
// Every kernel that uses SYNC() runs ONE wavefront per workgroup.  Lanes of a wave execute in lockstep and the LDS
// unit serves a wave's operations in issue order, so cross-lane exchange through LDS needs no s_barrier and no
// s_waitcnt in the HARDWARE -- only that the COMPILER keeps the program order of the LDS accesses around the exchange point.
// SYNC() = wavefront-scope fence + llvm.amdgcn.wave.barrier: neither emits an instruction; the fence orders the accesses for the IR
// optimisers, the wave barrier is the scheduling barrier that the machine scheduler may move NOTHING across.
// [r5] Up to libcrx 0.2.1 SYNC() was the fence alone, and that is not enough: to the compiler the lanes are independent threads, so
// two LDS accesses of "one thread" to provably different addresses (store ga[lane], load Jc[(lane / 12) ...]) are reorderable, and
// with alias analysis in codegen the machine scheduler did reorder across a fence in one A/B build of round 4 (DESIGN.md section 8,
// observation (1): run-time-horizon instantiation <2,24,6,0>, iterative-ilp strategy -- the first 64 entries of the Lagrangian
// gradient reached the adjoint sweep stale; -amdgpu-use-aa-in-codegen=0, -enable-misched=0, the default strategy and this wave barrier
// each make that build pass, tools/bughunt_*.sh).  Measured: the barrier costs nothing (cfg2 0.471 -> 0.473 ms, cfg4 11.38 -> 11.41 ms,
// cfg3 0.275 -> 0.276 ms; the phases are LDS round trips the scheduler could not overlap anyway).
// CRX_SYNC_FENCE_ONLY restores the old definition for A/B runs, CRX_SYNC_BARRIER a real s_barrier.
#if defined(CRX_SYNC_BARRIER)
#define SYNC() __syncthreads()
#elif defined(CRX_SYNC_FENCE_ONLY)
#define SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#else
#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// (1) wave primitives: DPP butterflies inside each 16-lane row, then the four row totals are read
// as scalars -- ~10 VALU ops instead of the 12 ds_bpermute round trips __shfl_xor lowers to.
// Every lane returns the full-wave result (wave-uniform).
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_f64(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// [r4] acc += m[i] * (lane FIRST + i of x), i = 0 .. CNT-1 in that order, ONE instruction per term: gfx90a+ lets the FP64 ALU take a DPP operand
// with row_newbcast (lane n of each 16-lane row to every lane of the row), so the broadcast and the FMA are the same instruction --
// v_readlane x 2 + v_fma before (the adjoint and forward sweeps: 14 + 20 of them per stage).  Valid for the lanes of row 0 reading lanes
// < 16 (the sweeps live in lanes < NZ <= 16; the other rows compute on their own lanes and are discarded).  Same product, same
// accumulation order, one rounding: identical bits.  The hazard recogniser does not look into inline assembly, and a DPP read of a VGPR
// needs two wait states after a VALU instruction that wrote it: the whole dot product is ONE asm statement that opens with s_nop 1, so
// that neither the producer of x nor a copy the register allocator may insert can sit closer than that; and acc is an EARLY-CLOBBER operand
// (+&v): were it given the register of x (same value at entry, as in `acc = row_dot(acc_old, m, acc_old)`), the second term would read through
// DPP what the first has just written -- the same hazard inside the statement (seen: 87 % of a 4096-problem batch lost).  EXEC covers the lanes read.
#define CRX_DPPT(i) "v_fmac_f64_dpp %0, %1, %" #i " row_newbcast:%"
#define CRX_DPPE " row_mask:0xf bank_mask:0xf\n\t"
template <int CNT, int FIRST>
__device__ __forceinline__ double row_dot(double x, const double* m, double acc) {
    static_assert(CNT >= 2 && CNT <= 9 && FIRST >= 0 && FIRST + CNT <= 16, "row_newbcast reaches the 16 lanes of a row");
    if constexpr (CNT == 2) asm("s_nop 1\n\t" CRX_DPPT(2) "4" CRX_DPPE CRX_DPPT(3) "5" CRX_DPPE : "+&v"(acc) : "v"(x), "v"(m[0]), "v"(m[1]), "n"(FIRST + 0), "n"(FIRST + 1));
    else if constexpr (CNT == 3) asm("s_nop 1\n\t" CRX_DPPT(2) "5" CRX_DPPE CRX_DPPT(3) "6" CRX_DPPE CRX_DPPT(4) "7" CRX_DPPE : "+&v"(acc) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "n"(FIRST + 0), "n"(FIRST + 1), "n"(FIRST + 2));
    else if constexpr (CNT == 4) asm("s_nop 1\n\t" CRX_DPPT(2) "6" CRX_DPPE CRX_DPPT(3) "7" CRX_DPPE CRX_DPPT(4) "8" CRX_DPPE CRX_DPPT(5) "9" CRX_DPPE : "+&v"(acc) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "n"(FIRST + 0), "n"(FIRST + 1), "n"(FIRST + 2), "n"(FIRST + 3));
    else if constexpr (CNT == 5) asm("s_nop 1\n\t" CRX_DPPT(2) "7" CRX_DPPE CRX_DPPT(3) "8" CRX_DPPE CRX_DPPT(4) "9" CRX_DPPE CRX_DPPT(5) "10" CRX_DPPE CRX_DPPT(6) "11" CRX_DPPE : "+&v"(acc) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "n"(FIRST + 0), "n"(FIRST + 1), "n"(FIRST + 2), "n"(FIRST + 3), "n"(FIRST + 4));
    else if constexpr (CNT == 6) asm("s_nop 1\n\t" CRX_DPPT(2) "8" CRX_DPPE CRX_DPPT(3) "9" CRX_DPPE CRX_DPPT(4) "10" CRX_DPPE CRX_DPPT(5) "11" CRX_DPPE CRX_DPPT(6) "12" CRX_DPPE CRX_DPPT(7) "13" CRX_DPPE : "+&v"(acc) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "n"(FIRST + 0), "n"(FIRST + 1), "n"(FIRST + 2), "n"(FIRST + 3), "n"(FIRST + 4), "n"(FIRST + 5));
    else if constexpr (CNT == 7) asm("s_nop 1\n\t" CRX_DPPT(2) "9" CRX_DPPE CRX_DPPT(3) "10" CRX_DPPE CRX_DPPT(4) "11" CRX_DPPE CRX_DPPT(5) "12" CRX_DPPE CRX_DPPT(6) "13" CRX_DPPE CRX_DPPT(7) "14" CRX_DPPE CRX_DPPT(8) "15" CRX_DPPE : "+&v"(acc) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "n"(FIRST + 0), "n"(FIRST + 1), "n"(FIRST + 2), "n"(FIRST + 3), "n"(FIRST + 4), "n"(FIRST + 5), "n"(FIRST + 6));
    else if constexpr (CNT == 8) asm("s_nop 1\n\t" CRX_DPPT(2) "10" CRX_DPPE CRX_DPPT(3) "11" CRX_DPPE CRX_DPPT(4) "12" CRX_DPPE CRX_DPPT(5) "13" CRX_DPPE CRX_DPPT(6) "14" CRX_DPPE CRX_DPPT(7) "15" CRX_DPPE CRX_DPPT(8) "16" CRX_DPPE CRX_DPPT(9) "17" CRX_DPPE : "+&v"(acc) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "n"(FIRST + 0), "n"(FIRST + 1), "n"(FIRST + 2), "n"(FIRST + 3), "n"(FIRST + 4), "n"(FIRST + 5), "n"(FIRST + 6), "n"(FIRST + 7));
    else if constexpr (CNT == 9) asm("s_nop 1\n\t" CRX_DPPT(2) "11" CRX_DPPE CRX_DPPT(3) "12" CRX_DPPE CRX_DPPT(4) "13" CRX_DPPE CRX_DPPT(5) "14" CRX_DPPE CRX_DPPT(6) "15" CRX_DPPE CRX_DPPT(7) "16" CRX_DPPE CRX_DPPT(8) "17" CRX_DPPE CRX_DPPT(9) "18" CRX_DPPE CRX_DPPT(10) "19" CRX_DPPE : "+&v"(acc) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "n"(FIRST + 0), "n"(FIRST + 1), "n"(FIRST + 2), "n"(FIRST + 3), "n"(FIRST + 4), "n"(FIRST + 5), "n"(FIRST + 6), "n"(FIRST + 7), "n"(FIRST + 8));
    return acc;
}

// [r6] The same inside each 8-LANE HALF of a row: returns, in lane 8 g + c, sum_{j < 6} m[j] * x(lane 8 g + j), j ascending, from 0 -- the T phase
// of the Riccati sweep with P in registers (lane 8 i + j holds P[i][j]: riccati_backward, PTR).  row_newbcast reaches one lane per 16-lane row, so
// TWO accumulators run over all lanes -- lane j of the row (right for lanes 0..7 of the row) and lane 8 + j (right for lanes 8..15) -- and the caller's
// `upper` (lane & 8) selects.  The two chains are independent and interleave: 12 issue slots for what was 6 LDS reads + 6 FMAs behind a
// store -> load round trip.  (bank_mask does NOT do this with one accumulator: on gfx950 a v_fmac_f64_dpp lane whose bank is masked off comes back
// ZERO, not unchanged -- tools/ubench/halfrow.hip.)  s_nop 1 and the early-clobber accumulators: as in row_dot.
__device__ __forceinline__ double halfrow_dot6(double x, const double* m, bool upper) {
    double lo = 0.0, hi = 0.0;
#define CRX_HR(i, j, j8) "v_fmac_f64_dpp %0, %2, %" #i " row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %" #i " row_newbcast:" #j8 " row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t" CRX_HR(3, 0, 8) CRX_HR(4, 1, 9) CRX_HR(5, 2, 10) CRX_HR(6, 3, 11) CRX_HR(7, 4, 12) CRX_HR(8, 5, 13)
        : "+&v"(lo), "+&v"(hi) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]));
#undef CRX_HR
    return upper ? hi : lo;
}

#define ROW_REDUCE(v, OP)                                                         \
    v = OP(v, dpp_f64<0xB1>(v));  /* quad_perm [1,0,3,2] */                       \
    v = OP(v, dpp_f64<0x4E>(v));  /* quad_perm [2,3,0,1] */                       \
    v = OP(v, dpp_f64<0x141>(v)); /* row_half_mirror     */                       \
    v = OP(v, dpp_f64<0x140>(v)); /* row_mirror          */
__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
__device__ __forceinline__ double op_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double op_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double op_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double wave_sum(double v) {
    ROW_REDUCE(v, op_add)
    return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}
__device__ __forceinline__ double wave_prod(double v) {
    ROW_REDUCE(v, op_mul)
    return (lane_f64(v, 0) * lane_f64(v, 16)) * (lane_f64(v, 32) * lane_f64(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
    ROW_REDUCE(v, op_max)
    return fmax(fmax(lane_f64(v, 0), lane_f64(v, 16)), fmax(lane_f64(v, 32), lane_f64(v, 48)));
}
__device__ __forceinline__ double wave_min(double v) {
    ROW_REDUCE(v, op_min)
    return fmin(fmin(lane_f64(v, 0), lane_f64(v, 16)), fmin(lane_f64(v, 32), lane_f64(v, 48)));
}

// Packed reductions [r2]: two or four wave reductions for little more than the price of one.  A single f64 reduction
// costs 23 VALU instructions, 12 of them the four DPP steps inside the 16-lane rows; the interior-point iteration does
// 17 of them, ~15 % of its VALU instructions.  gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves /
// alternate rows between TWO registers in one instruction, which lets several values share those four steps:
//   swap32(a, b): a' = (a.lo32, b.lo32), b' = (a.hi32, b.hi32);  a' OP b' = rows [A A B B] (lane partials of A in rows
//   0-1, of B in rows 2-3);  swap16(s1, s2): s1' = rows [s1.0 s2.0 s1.2 s2.2], s2' = rows [s1.1 s2.1 s1.3 s2.3];
//   s1' OP s2' = rows [A C B D]: ONE row reduction, four readlanes.  29 instructions for four values (92), 24 for two (46).
// All 64 lanes must be active (they are: the callers sit in wave-uniform control flow).
// gfx950 (MI355X, CDNA4) ONLY: these two instructions do not exist on gfx942 and older.  libcrx is written for this one
// target (no multi-arch fallback paths by design); another --offload-arch stops here with a readable message instead of
// an "unknown builtin" deep inside the solver.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libcrx targets gfx950 (MI355X / CDNA4) only: crx_wave.h uses v_permlane32_swap / v_permlane16_swap (build with ARCH=gfx950)"
#endif
__device__ __forceinline__ void swap32_f64(double& x, double& y) {
    auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    x = __hiloint2double(hi[0], lo[0]);
    y = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap16_f64(double& x, double& y) {
    auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    x = __hiloint2double(hi[0], lo[0]);
    y = __hiloint2double(hi[1], lo[1]);
}
#define WAVE_REDUCE4(a, b, c, d, OP)                                              \
    do {                                                                          \
        swap32_f64(a, b); double s1_ = OP(a, b);                                  \
        swap32_f64(c, d); double s2_ = OP(c, d);                                  \
        swap16_f64(s1_, s2_); double t_ = OP(s1_, s2_);                           \
        ROW_REDUCE(t_, OP)                                                        \
        a = lane_f64(t_, 0); c = lane_f64(t_, 16); b = lane_f64(t_, 32); d = lane_f64(t_, 48); \
    } while (0)
#define WAVE_REDUCE2(a, b, OP)                                                    \
    do {                                                                          \
        swap32_f64(a, b); double s1_ = OP(a, b), s2_ = s1_;                       \
        swap16_f64(s1_, s2_); double t_ = OP(s1_, s2_);                           \
        ROW_REDUCE(t_, OP)                                                        \
        a = lane_f64(t_, 0); b = lane_f64(t_, 32);                                \
    } while (0)
__device__ __forceinline__ void wave_sum4(double& a, double& b, double& c, double& d) { WAVE_REDUCE4(a, b, c, d, op_add); }
__device__ __forceinline__ void wave_max4(double& a, double& b, double& c, double& d) { WAVE_REDUCE4(a, b, c, d, op_max); }
__device__ __forceinline__ void wave_sum2(double& a, double& b) { WAVE_REDUCE2(a, b, op_add); }
__device__ __forceinline__ void wave_max2(double& a, double& b) { WAVE_REDUCE2(a, b, op_max); }

// sum over the wave of log(v), v > 0: mantissas multiplied, exponents added, ONE log per wave
// (a product of <= 6*64 mantissas in [0.5,1) cannot underflow: 2^-384)
struct LogAcc {
    double m;
    int e;
    __device__ __forceinline__ LogAcc() : m(1.0), e(0) {}
    __device__ __forceinline__ void mul(double v) {
        int ex;
        m *= frexp(v, &ex);
        e += ex;
    }
    __device__ __forceinline__ double wave_total() {
        return log(wave_prod(m)) + 0.6931471805599453 * wave_sum((double)e);
    }
    // the same with the exponent sum sharing its reduction with another sum of the caller (`other`, reduced in place)
    __device__ __forceinline__ double wave_total_with(double& other) {
        double ex = (double)e;
        wave_sum2(other, ex);
        return log(wave_prod(m)) + 0.6931471805599453 * ex;
    }
};

// select with BOTH operands evaluated first (function arguments).  `c ? expr : 0.0` with arithmetic or a load in an arm
// is emitted as control flow, and the optimiser then sinks the arm's loads into it: a predicated region (exec-mask
// traffic) plus one more serialized LDS round trip.  sel()/seli() keep it a v_cndmask.
__device__ __forceinline__ double sel(bool c, double a, double b) { return c ? a : b; }
__device__ __forceinline__ int seli(bool c, int a, int b) { return c ? a : b; }

// log2(x), x >= 0, to ~1e-7 absolute over the whole double range: exponent from frexp, mantissa through the
// single-precision hardware log (v_log_f32).  log2_fast(0) = -inf.
__device__ __forceinline__ double log2_fast(double x) {
    int e;
    const double m = frexp(x, &e);
    return (double)e + (double)__builtin_amdgcn_logf((float)m);
}

// 1/x to ~1 ulp: hardware estimate (v_rcp_f64) + two Newton steps; ~5 dependent ops instead of the
// ~10 of an IEEE division.  x must be finite, normal and non-zero (true for pivots and slacks).
__device__ __forceinline__ double frcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

// 1/sqrt(x) to ~1 ulp: v_rsq_f64 + two Newton steps.  x finite, normal, > 0.
__device__ __forceinline__ double frsqrt(double x) {
    const double hx = -0.5 * x;           // off the dependent chain: three dependent ops per Newton step instead of four
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(hx * r, r, 1.5);
    r = r * fma(hx * r, r, 1.5);
    return r;
}

// Wave-wide exclusive prefix / suffix sums, lane i = element i: Hillis-Steele inside each 16-lane row with DPP row
// shifts (zero fill), the three row totals added as scalars; the exclusive forms shift the input by one lane first
// (wave_shr:1 / wave_shl:1) instead of subtracting the own element from the inclusive sum (no cancellation).
// ~190 cycles as a dependent link (tools/ubench/scan.hip).
template <int CTRL>
__device__ __forceinline__ double dpp_zfill_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double excl_prefix(double v, int lane) {
    v = dpp_zfill_f64<0x138>(v);       // wave_shr:1
    v += dpp_zfill_f64<0x111>(v);      // row_shr:1
    v += dpp_zfill_f64<0x112>(v);      // row_shr:2
    v += dpp_zfill_f64<0x114>(v);      // row_shr:4
    v += dpp_zfill_f64<0x118>(v);      // row_shr:8
    const double r0 = lane_f64(v, 15), r1 = lane_f64(v, 31), r2 = lane_f64(v, 47);
    const int row = lane >> 4;
    return v + sel(row >= 1, r0, 0.0) + sel(row >= 2, r1, 0.0) + sel(row >= 3, r2, 0.0);
}
__device__ __forceinline__ double excl_suffix(double v, int lane) {
    v = dpp_zfill_f64<0x130>(v);       // wave_shl:1
    v += dpp_zfill_f64<0x101>(v);      // row_shl:1
    v += dpp_zfill_f64<0x102>(v);
    v += dpp_zfill_f64<0x104>(v);
    v += dpp_zfill_f64<0x108>(v);
    const double r1 = lane_f64(v, 16), r2 = lane_f64(v, 32), r3 = lane_f64(v, 48);
    const int row = lane >> 4;
    return v + sel(row <= 2, r3, 0.0) + sel(row <= 1, r2, 0.0) + sel(row <= 0, r1, 0.0);
}

// ------------------------------------------------------------------------------------------------
// dense factorisations in LDS, one matrix row per lane (crx_lmpc.hip, crx_path kernel in crx_prep.hip)
// ------------------------------------------------------------------------------------------------
// left-looking Cholesky, in place, of the leading n x n block (lower triangle) of a row-major array
// (stride LD); rows n..n+extra-1 are carried along, i.e. forward-substituted right-hand sides.
// Lane i owns row i.  Blocked by 4 columns: the panel product against all finished columns is one long
// loop of independent LDS reads (own row entry + 4 broadcast entries per k), the 4x4 diagonal block is
// finished in registers with v_readlane broadcasts -- one LDS round trip per 4 columns instead of per
// column.  Returns 0 if a pivot is not positive.  The inverse pivots go to sm[inv + j * inv_st] (inv_st = LD: a column of
// the array itself, e.g. its padding column from row 1 on -- row 0's last column is the sink of the masked stores).
__device__ __forceinline__ int l_chol(double* sm, int base, int LD, int inv, int n, int extra, int lane, int inv_st = 1) {
    const int rows = n + extra;
    const int ri = base + lane * LD;
    int ok = 1;
    for (int j0 = 0; j0 < n; j0 += 4) {
        const bool mine = lane >= j0 && lane < rows;
        // lanes without a row in this panel run the same instructions on row j0 (valid data) and store nothing:
        // no predicated region around the panel loop
        const int rr = mine ? ri : base + j0 * LD;
        double s[4], l[4];
        const int r0 = base + j0 * LD;
#pragma unroll
        for (int c = 0; c < 4; c++) s[c] = sm[rr + j0 + c];
        // panel product against the finished columns, four at a time (j0 is a multiple of 4): 20 LDS operands are
        // loaded back to back, then 16 FMAs on four independent accumulators
        for (int k = 0; k < j0; k += 4) {
            double a[4], b[4][4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) a[kk] = sm[rr + k + kk];
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int kk = 0; kk < 4; kk++) b[c][kk] = sm[r0 + c * LD + k + kk];   // rows j0+c <= n+2: inside the array
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
#pragma unroll
                for (int c = 0; c < 4; c++) s[c] = fma(-a[kk], b[c][kk], s[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = j0 + c;
            const bool col = j < n;               // wave-uniform
            const int jc = col ? j : n - 1;
#pragma unroll
            for (int cc = 0; cc < c; cc++) s[c] = fma(-l[cc], lane_f64(l[cc], jc), s[c]);
            const double d = lane_f64(s[c], jc);
            if (col && !(d > 0.0)) ok = 0;
            const double rinv = frsqrt(d);
            l[c] = sel(col, sel(lane == j, d, s[c]) * rinv, 0.0);
            const int sink = base + LD - 1;       // last column of row 0: upper triangle, never read
            sm[seli(col && mine && lane >= j, ri + j, sink)] = l[c];
            sm[seli(lane == 0 && col, inv + j * inv_st, sink)] = rinv;
        }
        if (!ok) return 0;
        SYNC();
    }
    return 1;
}

// x <- L^-T x for `NR` right-hand sides held one entry per lane (lane i = entry i), L as above.  The
// LDS operands of four steps are fetched ahead of the dependent readlane/fma chain.
template <int NR>
__device__ __forceinline__ void l_backsub(const double* sm, int base, int LD, int inv, int n, int lane, double* b, int inv_st = 1) {
    int j = n - 1;
    for (; j >= 3; j -= 4) {
        double rinv[4], lj[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            rinv[q] = sm[inv + (j - q) * inv_st];
            lj[q] = sel(lane < j - q, sm[base + (j - q) * LD + seli(lane < j - q, lane, 0)], 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const double xj = lane_f64(b[r], j - q) * rinv[q];
                b[r] = sel(lane == j - q, xj, fma(-lj[q], xj, b[r]));      // lj = 0 on the pivot lane
            }
    }
    for (; j >= 0; j--) {
        const double rinv = sm[inv + j * inv_st];
        const double lj = sel(lane < j, sm[base + j * LD + seli(lane < j, lane, 0)], 0.0);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const double xj = lane_f64(b[r], j) * rinv;
            b[r] = sel(lane == j, xj, fma(-lj, xj, b[r]));
        }
    }
}

#endif
