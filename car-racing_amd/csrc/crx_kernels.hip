// crx_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of libcrx.
//
// One finite-horizon optimal-control problem per 64-lane wavefront (one single-wave workgroup per
// problem).  The whole interior-point solve -- iterate, slacks, multipliers, filter, Riccati
// factors -- lives in that workgroup's LDS slice; HBM is touched twice: one coalesced read of the
// problem's inputs, one coalesced write of its trajectory.  FP64 throughout (the reduced Hessians
// have condition numbers 1e6..1e8 and the degree-6 barrier rows reach 1e10).
// No MFMA: the largest dense block is 9x14.
//
// Algorithm (the same mathematical iteration as oracle/crx_oracle.c, which
// factorises the condensed Newton system with a dense Cholesky instead):
//   primal-dual interior point on  min f(z)  s.t. c_j(z) - t_j = 0, t_j >= 0  for every inequality,
//   dynamics kept satisfied exactly (linear, x0 fixed), Newton step by a Riccati recursion over the
//   augmented stage state (x_k, sigma_k) / input (u_k, sigma_{k+1}), inertia correction through the
//   Riccati pivots, monotone barrier update, fraction-to-the-boundary rule, filter line search.
//
// What each block restates (paths into /root/reference/car_racing):
//   planner region QP   planning/overtake_traj_planner.py:263-334, fall-back :365-374
//   MPC-CBF NLP         control/control.py:492-591 (mpccbf), :270-382 (mpc_multi_agents)
//   region selection    planning/overtake_traj_planner.py:205-246
//
// Structure of this file: (1) wave primitives, (2) compile-time LDS layout, (3) set-up of the
// canonical stage problem, (4) per-iteration phases, (5) the solver kernel, (6) selection kernel,
// (7) launchers.  Stage coordinates everywhere:  z_k = [x_k (6), sigma_k (NOBS), u_k (2),
// sigma_{k+1} (NOBS)];  the iterate is stored as Z[k][NZ], the Newton step as dZ[k][NZ]
// (sigma_{k+1} therefore appears twice, as an input of stage k and as a state of stage k+1; the
// two copies are kept identical).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "crx_kparams.h"
#include "crx_wave.h"

#ifndef CRX_KKT_DIAG
#define CRX_KKT_DIAG 1   /* 0: without the unscaled-KKT diagnostics block of the write-back (A/B builds) */
#endif
#ifndef CRX_TU_GENERAL
#define CRX_TU_GENERAL 0   /* 1: crx_kernels_gen.hip -- the general instantiations, built conservatively (section (7)) */
#endif
#ifndef CRX_OPAQUE_LANE
#define CRX_OPAQUE_LANE 1 /* make EXTRA=-DCRX_OPAQUE_LANE=0: round-2 behaviour (lane maps hoisted out of the interior-point loop) */
#endif
#ifndef CRX_OPAQUE_OUTER
#define CRX_OPAQUE_OUTER 0 /* 0: A/B builds without the per-pass lane barrier of the (re)start loop */
#endif
#ifndef CRX_W2_FLOOR
#define CRX_W2_FLOOR 0 /* 1: pin <2,12> at two waves per SIMD (256 registers, 76 B of scratch) = 6 instead of 4 problems per CU */
#endif
#ifndef CRX_STATIC_LDS
#define CRX_STATIC_LDS 1   // 0: the solver's LDS as a dynamic (extern) array, as up to libcrx 0.2.0 (A/B builds)
#endif
#ifndef CRX_SWEEP_UNROLL
#define CRX_SWEEP_UNROLL 1   // 1: the forward and adjoint sweeps of the fixed-horizon instantiations are unrolled completely; 0: two stages per trip
#endif
#ifndef CRX_SWEEP_MASK
#define CRX_SWEEP_MASK 1   // 1: the adjoint / forward sweeps run as one predicated region over lanes < NZ; 0: every lane runs them, sink stores (A/B builds)
#endif
#ifndef CRX_FWD_ONE_DOT
#define CRX_FWD_ONE_DOT 1   // forward sweep: one lane-specific dot product per lane and stage (0: state and input chains in every lane; A/B builds)
#endif
#ifndef CRX_SWEEP_LOCAL_LANE
// bit 0: forward sweep, bit 1: adjoint sweep.  The forward sweep alone removes the scratch frame.  (A build of a later source state with BOTH
// failed tests/test_gpu_parity.py::test_fuzz_descriptors on <2,24,6,0> -- e_d of the second iteration off by 3 % -- while either one alone, the
// unmasked sweeps, the v_readlane sweeps and a build with s_barrier at every SYNC passed: not understood)
#define CRX_SWEEP_LOCAL_LANE 1
#endif
#ifndef CRX_STAGE_FENCE
#define CRX_STAGE_FENCE 1
#endif
#ifndef CRX_RIC_UNROLL
#define CRX_RIC_UNROLL 0   // stages of the Riccati backward sweep per trip of its loop in the fixed-horizon instantiations; 0 = all of them
#endif
#ifndef CRX_NFIX
#define CRX_NFIX 1   // 0: every launch reads the horizon from its arguments (A/B builds)
#endif
#ifndef CRX_DEG6
#define CRX_DEG6 1   // 0: every CBF launch takes the general-exponent instantiation (A/B builds)
#endif
#ifndef CRX_T_PAD
#define CRX_T_PAD 0
#endif
#ifndef CRX_MASK_FMA
#define CRX_MASK_FMA 1   // [r6] Riccati sweep: "add x on the lanes of a set" as fma(mask, x, t) with a 0 / 1 lane mask kept in registers instead of v_cndmask pairs + add (0: selects; A/B builds)
#endif
#ifndef CRX_H_MIRROR
#define CRX_H_MIRROR 0   // 1: the H phase of the Riccati sweep stores the upper triangle of H as well, as up to round 6 (A/B builds)
#endif
#ifndef CRX_HUU_FIRST
#define CRX_HUU_FIRST 1   // [r6] update phase of the Riccati sweep: the loads of Huu are issued before the other operands (0: the scheduler's order; A/B builds)
#endif
#ifndef CRX_PT_REG
#define CRX_PT_REG 1   // [r6] Riccati sweep of the 0- / 1-obstacle instantiations: (P | p) in registers, T by half-row broadcast-FMAs (0: P through LDS; A/B builds)
#endif
#ifndef CRX_SLIM
#define CRX_SLIM 1 /* make EXTRA=-DCRX_SLIM=0: the full LDS layout for every instantiation (A/B builds, tools/ab_slim.sh) */
#endif
#define MAXF 12 /* filter entries (reset at every barrier update; when full, further entries are dropped) */

// a^p for p in 0..8, straight-line: p is wave-uniform (the CBF degree 2/4/6/8, degree-1, degree-2), so every select
// below is a scalar-condition v_cndmask.  (A switch on p is a chain of scalar branches -- ~40 cycles and, worse, a
// basic-block boundary per factor inside the row passes.)
__device__ __forceinline__ double ipow_d(double a, int p) {
    const double a2 = a * a;
    const int h = p >> 1;
    double r = sel(p & 1, a, 1.0);
    r = sel(h >= 1, r * a2, r);
    r = sel(h >= 2, r * a2, r);
    r = sel(h >= 3, r * a2, r);
    r = sel(h >= 4, r * a2, r);
    // [r3] with a compile-time exponent (crx_solve_kernel<.., DEG = 6>) the chain above folds into plain products, and the last of
    // them would be CONTRACTED into the caller's addition (hipcc fuses across inlined code): different last bits than the select
    // chain gives, enough to move a chaotic closed loop (test_mpccbf_racing_m_shape).  The empty asm keeps the product a product.
    asm volatile("" : "+v"(r));
    return r;
}

// scipy interp1d(kind="linear") (searchsorted-left, index clipped to [1,n-1], slope form)
__device__ __forceinline__ double interp_lin(const double* xs, const double* ys, int n, double x) {
    int hi = 0;
    while (hi < n && xs[hi] < x) hi++;
    hi = hi < 1 ? 1 : (hi > n - 1 ? n - 1 : hi);
    int lo = hi - 1;
    double slope = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo]);
    return slope * (x - xs[lo]) + ys[lo];
}

// ------------------------------------------------------------------------------------------------
// (2) compile-time LDS layout.  NMAX bounds the horizon of an instantiation (12 or 24); every
// offset is a constant, so LDS addresses are immediates and no pointer lives in a register.
// Row slots inside a stage k: 0..3 input box (d lo, d hi, a lo, a hi); 4..7 box of x_{k+1} (vx lo,
// vx hi, ey lo, ey hi); 8+o: sigma_{k+1}^o >= 0; 8+NOBS+o: CBF row (k,o).  Rows N*NR+o: sigma_0^o.
// ------------------------------------------------------------------------------------------------
// All rows of a problem, pass by pass, fully unrolled and STRAIGHT-LINE: a lane past the last row recomputes row 0
// (jv = false).  Its stores of pure functions of other arrays rewrite row 0 with the value lane 0 wrote; read-modify-
// write stores and stores that depend on jv go to the sink (SINK), sums and products are masked with jv, maxima need
// nothing.  No predicated region: the LDS loads of every pass are in flight together, and the compiler cannot sink
// loads into a branch (each such region was one more serialized LDS round trip).
#define ROWS(j, jv, lane_, m_) _Pragma("unroll") for (int q_ = 0; q_ < (L::MR + WAVE - 1) / WAVE; q_++) \
    if (const bool jv = (lane_) + q_ * WAVE < (m_); true) if (const int j = jv ? (lane_) + q_ * WAVE : 0; true)

#define COORDS(e, ev, lane_, n_) _Pragma("unroll") for (int q_ = 0; q_ < (L::NV + WAVE - 1) / WAVE; q_++) \
    if (const bool ev = (lane_) + q_ * WAVE < (n_); true) if (const int e = ev ? (lane_) + q_ * WAVE : 0; true)

#define SINK(cond, off) seli((cond), (off), L::dmy)
// The CBF rows (k, ob) of a problem, one lane each (N * NOBS <= 72): j is the row index, ev false on lanes past the last
// row, which recompute row (0, 0).  The row passes skip these rows (ROW_IS_CBF) -- evaluating them there meant every
// lane of EVERY pass ran the degree-6 evaluation for the sake of a dozen rows.
#define CBF_ROWS(j, k, ob, ev, lane_, n_) _Pragma("unroll") for (int q_ = 0; q_ < (NMAX * NOBS + WAVE - 1) / WAVE; q_++) \
    if (const bool ev = (lane_) + q_ * WAVE < (n_) * NOBS; true) if (const int e_ = ev ? (lane_) + q_ * WAVE : 0; true) \
    if (const int k = e_ / L::NO; true) if (const int ob = e_ - k * L::NO; true) if (const int j = k * NR + 8 + NOBS + ob; true)
// in the planner instantiation (193 VGPRs, far from the 256 that would cost a resident wave) the row passes load all
// their operands before computing; with obstacles the 1-obstacle instantiation sits at 255 VGPRs and the scheduler is
// left alone
#define ROW_LOADS_DONE() do { if (NOBS == 0) __builtin_amdgcn_sched_barrier(0); } while (0)
#define ROW_IS_CBF(j, n_) (NOBS > 0 && (j) < (n_) * NR && (j) % NR >= 8 + NOBS)
// nothing moves across: placed after the loads of a phase so that they are issued back to back
#define LOADS_DONE() __builtin_amdgcn_sched_barrier(0)
// end of a stage of an UNROLLED sweep [r4]: without it the scheduler lifts the loads of all later stages to the top of the sweep and the
// registers they occupy are paid for in scratch (a wave writes its spills once: 100 B per lane were x8 the algorithmic bytes written of a cfg2 launch)
#if CRX_STAGE_FENCE
#define STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define STAGE_FENCE() ((void)0)
#endif

// row table entries (unsigned 16-bit): index into Z / dZ in bits 0..12 (<= 25 * 14 coordinates), flags above
#define RIV_SIMPLE (1 << 13) /* table-driven row that is present: c = +-(z[iv] - bound) */
#define RIV_NEG (1 << 14)
#define RIV_IDX(pk) ((pk) & 0x1FFF)
#define RIV_SGN(pk) (((pk) & RIV_SIMPLE) ? (((pk) & RIV_NEG) ? -1.0 : 1.0) : 0.0)
// the index tables behind the doubles: ints at si, 16-bit entries at SH16(si), row-of-coordinate entries (8 bits while the
// instantiation has <= 127 rows) at VROW(si)
#define SH16(si_) ((unsigned short*)((si_) + L::SH_OFF))
#define RIVT(si_, j) ((int)SH16(si_)[L::riv + (j)])
#define UPDP(si_, l_) ((int)SH16(si_)[L::updP + (l_)])
#define VROW(si_) ((typename L::vrow_t*)(SH16(si_) + L::END_S16))

template <int NOBS, int NMAX>
struct Lay {
    static constexpr int NX = 6 + NOBS, NU = 2 + NOBS, NZ = NX + NU, NR = 8 + 2 * NOBS;
    static constexpr int NO = NOBS ? NOBS : 1;
    static constexpr int MR = NMAX * NR + NOBS;      // row capacity
    static constexpr int NV = (NMAX + 1) * NZ;       // stage-coordinate capacity
    static constexpr int M = 0;                      // [NX][NZ]   [A B; 0 I] in stage coordinates
    static constexpr int Z = M + NX * NZ;            // [NMAX+1][NZ] iterate
    static constexpr int dZ = Z + NV;                // Newton step
    static constexpr int xr = dZ + NV;               // [NMAX+1][6] tracking references
    static constexpr int wc = xr + (NMAX + 1) * 6;   // [NMAX] coupling weight on (ey_{k+1}-ey_k)^2
    static constexpr int obs_s = wc + NMAX;          // [NOBS][NMAX+1]   (CBF-only arrays have size 0 in the planner instantiation)
    static constexpr int obs_e = obs_s + NOBS * (NMAX + 1);
    static constexpr int rt = obs_e + NOBS * (NMAX + 1);  // rows: slack, multiplier, value, steps, ...
    static constexpr int rnu = rt + MR;
    static constexpr int rc = rnu + MR;
    static constexpr int rdt = rc + MR;
    static constexpr int rtt = rdt + MR;             // trial slack / 1/t (dnu is recomputed at accept: -w + Sigma (rp - dt))
    // SLIM layout [r3] (the 3-obstacle instantiations: BASELINE configs[3], N = 20): four problems per CU need <= 40 960 B
    // and the full layout is 52 560 B.  Whatever is a pure function of other LDS contents is recomputed where it is used
    // -- Sigma and w from (t, nu, c, mu), the bound of a simple row from its slot, the presence of a simple row from its
    // table entry, the rows of a coordinate and the triangle map from arithmetic -- and the curvature table G shares the
    // storage of the feedback gains Kk (G lives from first_order to assemble_newton, Kk from the backward to the forward
    // sweep).  Same operations on the same operands in the same order: results identical to the full layout bit for bit.
    static constexpr bool SLIM = CRX_SLIM && NOBS == 3 && NMAX == 20;   // the other 3-obstacle instantiations are register-bound at 4 per CU either way
    static constexpr int MRS = SLIM ? 0 : MR;        // length of the per-row arrays a slim layout does without
    static constexpr int rsig = rtt + MR;            // Sigma = nu/t                               (full layout only)
    static constexpr int rw = rsig + MRS;            // w = nu - mu/t + Sigma*(c - t)              (full layout only)
    static constexpr int rsc = rw + MRS;             // row scale (0 = row absent)                 (full layout only)
    static constexpr int csc = rsc + MRS;            // slim: [NMAX][NO] scales of the CBF rows (0 = absent); simple rows: presence = RIV_SIMPLE
    static constexpr int rb = csc + (SLIM ? NMAX * NOBS : 0);   // simple rows: bound (their sign lives in the riv table)   (full layout only)
    static constexpr int Gpos = rb + MRS;            // [NMAX][NO][4] CBF curvatures at the iterate: d2/ds2, d2/dey2 of the "next" term, then
                                                     // of the "current" term (the gradients go straight into Jc); slim: over Kk, see G below
    static constexpr int Hd = Gpos + (SLIM ? 0 : NMAX * NOBS * 4);   // [NV] stage Hessian diagonal
    static constexpr int hg = Hd + NV;               // [NV] Newton gradient
    static constexpr int ga = hg;                    // Lagrangian gradient / reduced form: SAME storage -- assemble_newton turns ga[e]
                                                     // into hg[e] in place, and nothing reads ga again before first_order rebuilds it
    static constexpr int Jc = hg + NV;               // [NMAX][NO][NZ] CBF Jacobians (scaled)
    // [NMAX][2] "next" CBF curvature on (s_{k+1}, ey_{k+1}), interleaved and 16-byte aligned: the backward sweep reads the pair of a stage
    // with ONE ds_read_b128 [r4] (LDS instructions cost a lone wave more than their issue slot: the data path moves 128 B per clock)
    static constexpr int kS = (Jc + NMAX * NOBS * NZ + 1) & ~1;
    static constexpr int kE = kS + 1;
    // Riccati work.  The backward sweep runs while two arrays are dead: the row steps rdt (rewritten by the row-step pass
    // that follows the forward sweep) and the Newton step dZ beyond its first stage (rewritten by the forward sweep; the
    // first stage receives sigma_0 at the end of the backward sweep, after the last H is consumed).  P, pv, T live in the
    // former and H in the latter wherever they fit (they do for every instantiation but H of <3,12>; <2,12> fits exactly:
    // 28 412 -> 27 164 B; its residency stays at four per CU, the register file's limit: 296 registers, and a floor of two
    // waves per SIMD would park 92 of them).
    static constexpr int HS = NZ + 1;                // row stride of H: column NZ is the gradient hv (odd strides for NZ = 8, 10, 12, 14)
    // row stride of T (A/B builds, VERDICT r5 item 6: -DCRX_T_PAD=1 pads the even strides of the obstacle layouts to odd ones against LDS bank conflicts of
    // the T stores; measured round 6: profiles/r06_pmc_issue.txt)
    static constexpr int TS = NZ + ((CRX_T_PAD && NOBS > 0 && (NZ % 2 == 0)) ? 1 : 0);
    static constexpr bool PT_ALIAS = NX * NX + NX + NX * TS <= MR, H_ALIAS = NZ * HS <= NV;
    static constexpr int WORK = kS + (NOBS ? 2 * NMAX : 0);
    static constexpr int P = PT_ALIAS ? rdt : WORK;
    static constexpr int pv = P + NX * NX;
    static constexpr int T = pv + NX;
    static constexpr int WORK2 = PT_ALIAS ? WORK : T + NX * TS;
    static constexpr int H = H_ALIAS ? dZ : WORK2;           // [NZ][HS]  (over dZ from its start: stage 0 of dZ is written only after the sweep)
    static constexpr int Kk = H_ALIAS ? WORK2 : H + NZ * HS; // [NMAX][NU][NX]
    static constexpr int G = SLIM ? Kk : Gpos;
    // slim [r4b]: Sigma = nu / t of the CBF rows, [NMAX][NO], written by assemble_newton and read by the backward sweep's H phase -- behind H
    // in dZ, which is dead from the adjoint to the forward sweep (the full layout has rsig for this; recomputing it per stage and pass
    // was three LDS reads and three VALU instructions per obstacle: 360 of the 12.3 k instructions of a <3,20> iteration)
    static constexpr int sigS = (SLIM && H_ALIAS && NZ * HS + NMAX * NO <= NV) ? dZ + NZ * HS : -1;
    static_assert(NMAX * NOBS * 4 <= NMAX * NU * NX, "G fits inside Kk");
    static constexpr int kf = Kk + NMAX * NU * NX;   // [NMAX][NU]
    static constexpr int Fth = kf + NMAX * NU;
    static constexpr int Fph = Fth + MAXF;
    static constexpr int cst = Fph + MAXF;           // 0..5 wq, 6..7 wr, 8.. lap_off (NOBS <= 8), 16.. 1/l_sum per obstacle, 16+NOBS.. 1/w_sum per obstacle
    static constexpr int dmy = cst + 16 + 2 * NOBS;  // sink of the address-predicated stores (lanes without an entry write here)
    static constexpr int END_D = dmy + 2;
    // int tables (stored after the doubles)
    static constexpr int triH = 0;                   // [NZ(NZ+1)/2] packed (r << 8 | a) of the lower triangle of H
    static constexpr int END_I = triH + ((NZ * NZ <= WAVE || SLIM) ? 0 : NZ * (NZ + 1) / 2);   // slim: tri_decode()
    static constexpr int SH_OFF = (END_I + 1) & ~1;  // in ints, from si
    // 16-bit tables (after the ints; SH16())
    static constexpr int riv = 0;                    // [MR]  simple rows: index into Z / dZ | RIV_SIMPLE | RIV_NEG (sign of the Jacobian entry)
    static constexpr int updP = riv + MR;            // [64 * UCNT] packed (i << 8 | j) lane map of the Riccati update
    // passes of the update phase: NX (NX + 1) / 2 + NX entries of (P_new | p_new) + NX + 1 feedback columns, one lane each -- one pass for
    // NX <= 9 (up to three obstacles), two for the generic six-obstacle instantiation [r4]
    static constexpr int UCNT = (NX * (NX + 1) / 2 + NX + NX + 1 + WAVE - 1) / WAVE;
    static constexpr int END_S16 = (updP + 64 * UCNT + 1) & ~1;
    // row index of the lower- / upper-bound row of a coordinate (-1 none), after the 16-bit tables (VROW()): one byte
    // each while the row count allows -- with the 16-bit row table this takes the planner instantiation from 14 080 to
    // 13 552 B = 12 instead of 11 problems per CU (the runtime grants 12 up to 13 632 B: tools/ubench/lds_occupancy.hip)
    using vrow_t = std::conditional_t<(MR <= 127), signed char, short>;
    static constexpr int vlo = 0;                    // [NV]
    static constexpr int vhi = vlo + NV;             // [NV]
    static constexpr int END_V = SLIM ? 0 : vhi + NV;   // slim: coord_rows() computes them
    static constexpr size_t BYTES = (size_t)END_D * 8 + (size_t)SH_OFF * 4 + (size_t)END_S16 * 2 + (((size_t)END_V * sizeof(vrow_t) + 7) & ~(size_t)7);
    static_assert(!SLIM || BYTES <= 40960, "the slim layout exists to fit four problems per CU (160 KB / 4)");
    // [r6] SPECULATIVE second factorisation (crx_solve_kernel<.., SPEC = 1>, two waves per problem): everything riccati_backward WRITES, once more
    // behind the layout -- the second wave factorises the reduced Hessian with the NEXT entry of the inertia-correction schedule while the first
    // tries the current one.  In doubles from sm; ctl = {command, dw, convexified, ok} mailbox between the two waves.
    static constexpr int R2 = (int)((BYTES + 15) / 16) * 2;
    static constexpr int P2 = R2, pv2 = P2 + NX * NX, T2 = pv2 + NX, H2 = (T2 + NX * TS + 1) & ~1, Kk2 = H2 + NZ * HS, kf2 = Kk2 + NMAX * NU * NX,
                         dZ2 = kf2 + NMAX * NU, ctl = dZ2 + NZ, R2_END = ctl + 4;
    static constexpr size_t BYTES_SPEC = (size_t)R2_END * 8;
};

// problem context kept in registers (all wave-uniform)
struct Ctx {
    int N, lane, nobs, m;
    double lin_sN, cconst, wsig, alpha, om, cm;
    int degree;
    double b_d, b_a, b_vlo, b_vhi, b_e;   // delta_max, a_max, v_min, v_max, ey_max: the bounds of the simple rows (slim layout: row_bound())
};

// ---- accessors that hide the two layouts ---------------------------------------------------------------------------
// scale of row j (0 = row absent)
template <class L>
__device__ __forceinline__ double row_scale(const double* sm, const int* si, int j, int N) {
    if constexpr (L::SLIM) {
        const bool cbf = j < N * L::NR && j % L::NR >= 8 + L::NO;
        const int k = j / L::NR, ob = j - k * L::NR - 8 - L::NO;
        const double cs = sm[L::csc + seli(cbf, k * L::NO + ob, 0)];
        return sel(cbf, cs, sel((RIVT(si, j) & RIV_SIMPLE) != 0, 1.0, 0.0));
    } else {
        return sm[L::rsc + j];
    }
}
// scale of the CBF row (k, ob)
template <class L>
__device__ __forceinline__ double cbf_scale(const double* sm, int k, int ob) {
    if constexpr (L::SLIM) return sm[L::csc + k * L::NO + ob];
    else return sm[L::rsc + k * L::NR + 8 + L::NO + ob];
}
// bound of the simple row j (what set-up stores in rb: 0 for rows that are absent or not table-driven)
template <class L>
__device__ __forceinline__ double row_bound(const double* sm, const Ctx& c, int j) {
    if constexpr (L::SLIM) {
        const int r = j < c.N * L::NR ? j % L::NR : 8;
        const double ub = sel(r < 2, c.b_d, c.b_a) * sel((r & 1) != 0, 1.0, -1.0);
        const double xb = sel(r == 4, c.b_vlo, sel(r == 5, c.b_vhi, sel(r == 6, -c.b_e, c.b_e)));
        return sel(r < 4, ub, sel(r < 8, xb, 0.0));
    } else {
        return sm[L::rb + j];
    }
}
// Sigma = nu/t and w = nu - mu/t + Sigma (c - t) of row j: stored by assemble_newton (full layout) or recomputed from the
// row state with the same expressions (slim).  rti = 1/t as assemble_newton computed it (frcp(t)).
template <class L>
__device__ __forceinline__ void row_sig_w(const double* sm, int j, bool on, double mu, double t, double nu, double rti, double cj,
                                          double& sig, double& w) {
    if constexpr (L::SLIM) {
        const double sg = nu * rti;
        sig = sel(on, sg, 0.0);
        w = sel(on, nu - mu * rti + sg * (cj - t), 0.0);
    } else {
        sig = sm[L::rsig + j];
        w = sm[L::rw + j];
    }
}
// rows that bound stage coordinate e = (k, a) from below / above (-1: none)
template <class L>
__device__ __forceinline__ void coord_rows(const int* si, const Ctx& c, int e, int k, int a, int& rl, int& rh) {
    if constexpr (L::SLIM) {
        constexpr int NX = L::NX, NR = L::NR;
        const int N = c.N;
        const bool isx = a < 6, iss0 = a >= 6 && a < NX, isu = a >= NX && a < NX + 2;
        const bool xb = isx && k >= 1 && (a == 0 || a == 5);
        int jl = seli(xb, (k - 1) * NR + seli(a == 0, 4, 6), -1);
        jl = seli(iss0 && k == 0, N * NR + (a - 6), jl);
        jl = seli(isu && k < N, k * NR + 2 * (a - NX), jl);
        jl = seli(a >= NX + 2 && k < N, k * NR + 8 + (a - NX - 2), jl);
        const int jh = seli(xb || (isu && k < N), jl + 1, -1);
        // presence (absent obstacle slots, infinite bounds) is what the row table says
        rl = seli(jl >= 0 && (RIVT(si, seli(jl >= 0, jl, 0)) & RIV_SIMPLE) != 0, jl, -1);
        rh = seli(jh >= 0 && (RIVT(si, seli(jh >= 0, jh, 0)) & RIV_SIMPLE) != 0, jh, -1);
    } else {
        rl = VROW(si)[L::vlo + e];
        rh = VROW(si)[L::vhi + e];
    }
}
// entry e of the packed lower triangle -> (row, column)
__device__ __forceinline__ void tri_decode(int e, int& r, int& a) {
    int q = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    q += ((q + 1) * (q + 2) / 2 <= e) ? 1 : 0;
    q -= (q * (q + 1) / 2 > e) ? 1 : 0;
    r = q; a = e - q * (q + 1) / 2;
}

#define LD(off) sm[(off)]
#ifndef CRX_ROWDPP
#define CRX_ROWDPP 1   // 0: the sweeps broadcast with v_readlane (A/B builds)
#endif
// the adjoint / forward sweeps keep a stage's NZ numbers in lanes 0 .. NZ-1: inside one 16-lane row up to four obstacles
template <class L> constexpr bool ROWDPP = CRX_ROWDPP && L::NZ <= 16;

// phase clocks of the crx_trace_enable diagnostics.  s_memtime shares lgkmcnt with the LDS and returns out of order, so
// every clock read in a sweep turns the partial waits around it into full drains: compiled in only by `make TRACE=1`
#ifdef CRX_PHASE_CLOCKS
#define CLK() clock64()
#else
#define CLK() 0LL
#endif

// ------------------------------------------------------------------------------------------------
// (4) phases
// ------------------------------------------------------------------------------------------------
// CBF pieces of row (k,o) at Z + al*dZ: normalised distances to the obstacle at stages k and k+1
template <int NOBS, int NMAX>
__device__ __forceinline__ void cbf_dist(const double* sm, const Ctx& c, int k, int o, double al, double& dsc,
                                         double& dec, double& dsn, double& den) {
    using L = Lay<NOBS, NMAX>;
    const int N1 = c.N + 1;
    const double sc = LD(L::Z + k * L::NZ + 4) + al * LD(L::dZ + k * L::NZ + 4);
    const double ec = LD(L::Z + k * L::NZ + 5) + al * LD(L::dZ + k * L::NZ + 5);
    const double sn = LD(L::Z + (k + 1) * L::NZ + 4) + al * LD(L::dZ + (k + 1) * L::NZ + 4);
    const double en = LD(L::Z + (k + 1) * L::NZ + 5) + al * LD(L::dZ + (k + 1) * L::NZ + 5);
    // 1 / (l_agent + l_obs), 1 / (w_agent + w_obs) of THIS obstacle (control.py:529-535 takes them per obstacle)
    const double rLs = LD(L::cst + 16 + o), rWs = LD(L::cst + 16 + L::NO + o);
    dsc = (sc - LD(L::obs_s + o * N1 + k) - LD(L::cst + 8 + o)) * rLs;  // lap-corrected (control.py:539-540)
    dec = (ec - LD(L::obs_e + o * N1 + k)) * rWs;
    dsn = (sn - LD(L::obs_s + o * N1 + k + 1)) * rLs;                    // NOT corrected (control.py:542, quirk Q1)
    den = (en - LD(L::obs_e + o * N1 + k + 1)) * rWs;
}

template <int NOBS, int NMAX>
__device__ __forceinline__ double cbf_value(const double* sm, const Ctx& c, int k, int o, double al) {
    using L = Lay<NOBS, NMAX>;
    double dsc, dec, dsn, den;
    cbf_dist<NOBS, NMAX>(sm, c, k, o, al, dsc, dec, dsn, den);
    const int q = c.degree;
    const double gc = ipow_d(dsc, q) + ipow_d(dec, q), gn = ipow_d(dsn, q) + ipow_d(den, q);
    const double sk = LD(L::Z + k * L::NZ + 6 + o) + al * LD(L::dZ + k * L::NZ + 6 + o);
    const double sk1 = LD(L::Z + k * L::NZ + L::NX + 2 + o) + al * LD(L::dZ + k * L::NZ + L::NX + 2 + o);
    return gn - sk1 - c.om * (gc - sk) - c.alpha * c.cm;
}

// cost at Z + al*dZ (every lane returns the total)
template <int NOBS, int NMAX>
__device__ __forceinline__ double cost_value(const double* sm, const Ctx& c, double al) {
    using L = Lay<NOBS, NMAX>;
    const int N = c.N;
    double acc = 0.0;
    for (int e = c.lane; e < (N + 1) * 6; e += WAVE) {
        const int k = e / 6, i = e - k * 6;
        const double v = LD(L::Z + k * L::NZ + i) + al * LD(L::dZ + k * L::NZ + i);
        const double d = v - LD(L::xr + e);
        acc += LD(L::cst + i) * d * d;
        if (k == N && i == 4) acc += c.lin_sN * v;
    }
    for (int e = c.lane; e < N * 2; e += WAVE) {
        const int k = e >> 1, i = e & 1;
        const double v = LD(L::Z + k * L::NZ + L::NX + i) + al * LD(L::dZ + k * L::NZ + L::NX + i);
        acc += LD(L::cst + 6 + i) * v * v;
    }
    for (int k = c.lane; k < N; k += WAVE) {
        const double wck = LD(L::wc + k);
        const double e1 = LD(L::Z + (k + 1) * L::NZ + 5) + al * LD(L::dZ + (k + 1) * L::NZ + 5);
        const double e0 = LD(L::Z + k * L::NZ + 5) + al * LD(L::dZ + k * L::NZ + 5);
        acc += wck * (e1 - e0) * (e1 - e0);
    }
    if (NOBS) {
        for (int e = c.lane; e < (N + 1) * NOBS; e += WAVE) {
            const int k = e / L::NO, o = e - k * L::NO;
            if (o < c.nobs) acc += c.wsig * (LD(L::Z + k * L::NZ + 6 + o) + al * LD(L::dZ + k * L::NZ + 6 + o));
        }
    }
    return wave_sum(acc) + c.cconst;
}

// The cost is exactly quadratic along a step: f(Z + al dZ) = f + al (g'dZ) + al^2 (1/2 dZ'H dZ).  Returns the
// directional derivative g'dZ and the curvature term qq = 1/2 dZ'H dZ in one pass, so that a line-search
// trial costs no evaluation loop at all.
// (returns the LANE PARTIALS of both sums: the caller reduces them together with its own, wave_sum4)
template <int NOBS, int NMAX>
__device__ __forceinline__ double cost_dir(const double* sm, const Ctx& c, double& qq) {
    using L = Lay<NOBS, NMAX>;
    const int N = c.N;
    double acc = 0.0, q = 0.0;
    // one straight-line pass over the stage coordinates (the four loops of cost_value, by coordinate kind)
    COORDS(e, ev, c.lane, N * L::NZ + L::NX) {
        const int k = e / L::NZ, a = e - k * L::NZ;
        const bool isx = a < 6, isu = a >= L::NX && a < L::NX + 2, iss0 = a >= 6 && a < L::NX;
        const bool cpl = a == 5 && k >= 1;                                   // (ey_k - ey_{k-1})^2, weight wc[k-1]
        const double dv = LD(L::dZ + e), z = LD(L::Z + e);
        const double w = sel(isx || isu, LD(L::cst + (isx ? a : (isu ? 6 + a - L::NX : 0))), 0.0);
        const double ref = sel(isx, LD(L::xr + k * 6 + (isx ? a : 0)), 0.0);
        const int ep = seli(cpl, e - L::NZ, e);
        const double de = z - LD(L::Z + ep), dd = dv - LD(L::dZ + ep);
        const double wk = sel(cpl, LD(L::wc + seli(cpl, k - 1, 0)), 0.0);
        double t = 2.0 * (w * (z - ref) * dv + wk * de * dd);
        t += sel(k == N && a == 4, c.lin_sN * dv, 0.0);
        if (NOBS) t += sel(iss0 && a - 6 < c.nobs, c.wsig * dv, 0.0);
        acc += sel(ev, t, 0.0);
        q += sel(ev, w * dv * dv + wk * dd * dd, 0.0);
    }
    qq = q;
    return acc;
}

// row values at the iterate (al = 0): simple rows from the tables, CBF rows evaluated
template <int NOBS, int NMAX>
__device__ __forceinline__ void eval_rows(double* sm, const int* si, const Ctx& c) {
    using L = Lay<NOBS, NMAX>;
    for (int j = c.lane; j < c.m; j += WAVE) {
        const double sc = row_scale<L>(sm, si, j, c.N);
        const int pk = RIVT(si, j);
        double v = RIV_SGN(pk) * (LD(L::Z + RIV_IDX(pk)) - row_bound<L>(sm, c, j));
        if (NOBS && j < c.N * L::NR) {
            const int k = j / L::NR, r = j - k * L::NR;
            if (r >= 8 + NOBS && sc != 0.0) v = sc * cbf_value<NOBS, NMAX>(sm, c, k, r - 8 - NOBS, 0.0);
        }
        LD(L::rc + j) = (sc != 0.0) ? v : 1.0;
    }
}

// First-order pieces at the iterate: CBF derivative table G, CBF Jacobians Jc (scaled, stage
// coordinates) and the Lagrangian gradient ga = grad f - J' nu per stage coordinate.
template <int NOBS, int NMAX>
__device__ __forceinline__ void first_order(double* sm, const int* si, const Ctx& c) {
    using L = Lay<NOBS, NMAX>;
    const int N = c.N;
    if (NOBS) {
#pragma unroll
        for (int q_ = 0; q_ < (NMAX * NOBS + WAVE - 1) / WAVE; q_++) {
            const int e0 = c.lane + q_ * WAVE;
            const int e = e0 < N * NOBS ? e0 : 0;          // lanes past the table recompute entry 0 (same values)
            const int k = e / L::NO, o = e - k * L::NO;
            double* G = sm + L::G + (k * L::NO + o) * 4;
            double* J = sm + L::Jc + (k * L::NO + o) * L::NZ;
            // absent obstacle (o >= nobs): its data may be anything; every product below is discarded by a select, so
            // that 0 * (stale LDS) never becomes NaN in G / J
            const bool here = o < c.nobs;
            double dsc, dec, dsn, den;
            cbf_dist<NOBS, NMAX>(sm, c, k, o, 0.0, dsc, dec, dsn, den);
            const int q = c.degree;
            const double qd = (double)q, qq = (double)(q * (q - 1));
            const double p2sn = ipow_d(dsn, q - 2), p2en = ipow_d(den, q - 2);
            const double p2sc = ipow_d(dsc, q - 2), p2ec = ipow_d(dec, q - 2);
            const double rLs = LD(L::cst + 16 + o), rWs = LD(L::cst + 16 + L::NO + o);
            const double gsn = qd * p2sn * dsn * rLs, gen = qd * p2en * den * rWs;
            const double gsc = qd * p2sc * dsc * rLs, gec = qd * p2ec * dec * rWs;
            const double d = cbf_scale<L>(sm, k, o);
            double m4[L::NZ], m5[L::NZ];
#pragma unroll
            for (int a = 0; a < L::NZ; a++) { m4[a] = LD(L::M + 4 * L::NZ + a); m5[a] = LD(L::M + 5 * L::NZ + a); }
            G[0] = sel(here, qq * p2sn * rLs * rLs, 0.0); G[1] = sel(here, qq * p2en * rWs * rWs, 0.0);
            G[2] = sel(here, qq * p2sc * rLs * rLs, 0.0); G[3] = sel(here, qq * p2ec * rWs * rWs, 0.0);
            // every entry composed in registers and stored once (no read-modify-write round trips through LDS)
#pragma unroll
            for (int a = 0; a < L::NZ; a++) {
                double v = d * (gsn * m4[a] + gen * m5[a]);
                if (a == 4) v -= d * c.om * gsc;
                if (a == 5) v -= d * c.om * gec;
                v += sel(a == 6 + o, d * c.om, 0.0);
                v -= sel(a == L::NX + 2 + o, d, 0.0);
                J[a] = sel(here, v, 0.0);
            }
        }
        SYNC();
    }
    COORDS(e, ev, c.lane, N * L::NZ + L::NX) {   // stage N has states only (127 entries at N=12, 1 obstacle: two passes, not three)
        (void)ev;
        // selects instead of branches: every divergent region costs ~30 cycles on a lone wave
        const int k = e / L::NZ, a = e - k * L::NZ;
        const bool isx = a < 6, isu = a >= L::NX && a < L::NX + 2, iss0 = a >= 6 && a < L::NX, iss1 = a >= L::NX + 2;
        const int o = iss0 ? a - 6 : (iss1 ? a - L::NX - 2 : 0);
        const bool sig_on = (o < c.nobs) && ((iss0 && k == 0) || (iss1 && k < N));
        // every load is unconditional on a clamped index and selected afterwards: `cond ? LD(..) : 0` compiles to a
        // branch around the load (LDS loads are not speculated), i.e. one more serialized LDS round trip each
        const double cw = LD(L::cst + (isx ? a : (isu ? 6 + a - L::NX : 0))), xrv = LD(L::xr + k * 6 + (isx ? a : 0));
        const double w2 = sel(isx || isu, 2.0 * cw, 0.0);
        const double ref = sel(isx, xrv, 0.0);
        double g = w2 * (LD(L::Z + e) - ref);
        g += (k == N && a == 4) ? c.lin_sN : 0.0;
        g += sig_on ? c.wsig : 0.0;
        int rl, rh;
        coord_rows<L>(si, c, e, k, a, rl, rh);
        const double nl = LD(L::rnu + (rl >= 0 ? rl : 0)), nh = LD(L::rnu + (rh >= 0 ? rh : 0));
        g -= sel(rl >= 0, nl, 0.0);
        g += sel(rh >= 0, nh, 0.0);
        const int kk = k < N ? k : N - 1;                      // stage terms exist for k < N only
        if (NOBS == 0) {
            const double de = LD(L::Z + (kk + 1) * L::NZ + 5) - LD(L::Z + kk * L::NZ + 5);
            const double wck = LD(L::wc + kk), m5 = LD(L::M + 5 * L::NZ + a);
            g += sel(k < N, 2.0 * wck * de * (m5 - (a == 5 ? 1.0 : 0.0)), 0.0);
        } else {
#pragma unroll
            for (int ob = 0; ob < NOBS; ob++) {
                const double nuv = LD(L::rnu + kk * L::NR + 8 + NOBS + ob), jv2 = LD(L::Jc + (kk * L::NO + ob) * L::NZ + a);
                g -= sel(k < N, nuv * jv2, 0.0);
            }
        }
        g = (k == N && a >= L::NX) ? 0.0 : g;
        LD(L::ga + e) = g;
    }
    SYNC();
}

// Adjoint sweep.  Returns the infinity norm of the reduced Lagrangian gradient (inputs of every
// stage + sigma_0) and overwrites ga with the same gradient in "reduced form": the costates are
// folded into the stage gradients (adding lam'(M dz_k - dx_{k+1}) = 0 to the Newton QP), so the
// state part vanishes identically and the input part is the (small) reduced gradient.  The Riccati
// vector recursion then carries residual-sized numbers instead of O(nu) terms that cancel.
template <int NOBS, int NMAX, int UNR = 1>
__device__ __forceinline__ double dual_infeasibility(double* sm, const Ctx& c) {
    using L = Lay<NOBS, NMAX>;
    const int N = c.N;
    int lane = c.lane;
#if CRX_SWEEP_LOCAL_LANE & 2
    if (NOBS > 0) asm volatile("" : "+v"(lane));   // see riccati_forward
#endif
    double emax = 0.0;
    // The costate recursion lives in registers: lane i < NX carries lam[i], the NX values a stage needs are broadcast
    // with v_readlane (scalar operands of the FMAs) -- no LDS round trip on the dependent chain (it was one per
    // stage, ~350 cycles of 12); ga[k] does not depend on the chain, so its loads run ahead.
    // [r4] The sweep needs lanes 0 .. NZ-1 only: it runs as ONE predicated region (EXEC = those lanes; the DPP broadcasts and
    // v_readlane read lanes < NX, all inside), so that loads and stores address `base + lane` with the stage offset as an immediate --
    // sink-address selects, one per stage, were hoisted out of the interior-point loop and paid for in registers (scratch) once the
    // sweep was unrolled.
    double tot = 0.0;
    if (!CRX_SWEEP_MASK || lane < L::NZ) {
    constexpr bool MK = CRX_SWEEP_MASK;
    const bool in = MK || lane < L::NZ;
    const int la = in ? lane : 0;
    double mcol[L::NX];   // column `lane` of the (stage-invariant) model matrix, kept in registers across the sweep
#pragma unroll
    for (int i = 0; i < L::NX; i++) mcol[i] = LD(L::M + i * L::NZ + la);
    tot = LD(L::ga + N * L::NZ + la);               // lam_N = the terminal gradient (lanes < NX)
    double gk = LD(L::ga + (N - 1) * L::NZ + la);
    LD(SINK(in, L::ga + N * L::NZ + lane)) = 0.0;
    auto stage = [&](int k) {
        const double gn = LD(L::ga + (k >= 1 ? k - 1 : 0) * L::NZ + la);   // next stage's gradient, in flight during this one
        double t = gk;
        if constexpr (ROWDPP<L>) {
            t = row_dot<L::NX, 0>(tot, mcol, t);
        } else {
#pragma unroll
            for (int i = 0; i < L::NX; i++) t += mcol[i] * lane_f64(tot, i);
        }
        tot = t;
        emax = fmax(emax, sel(lane >= L::NX && in, fabs(tot), 0.0));
        const bool keep = lane >= L::NX || (k == 0 && lane >= 6);
        LD(SINK(in, L::ga + k * L::NZ + lane)) = sel(keep, tot, 0.0);
        gk = gn;
        STAGE_FENCE();
    };
    if constexpr (UNR > 1) {   // fixed horizon: straight-line code, stage addresses are immediates [r4]
#pragma unroll UNR
        for (int k = N - 1; k >= 0; k--) stage(k);
    } else {
        for (int k = N - 1; k >= 0; k--) stage(k);
    }
    }
    SYNC();
    if (NOBS && lane >= 6 && lane < 6 + c.nobs) emax = fmax(emax, fabs(tot));
    return wave_max(emax);
}

// Infeasibility certificate for the instantiations whose rows are all linear (NOBS == 0: planner region QPs, 0-obstacle
// NLPs) [r2]; same test as oracle/crx_oracle.c box_certificate().  With y >= 0 on the state rows c_j = +-(x_ki - bound),
// S(v) = sum_j y_j c_j(v) is linear in the inputs and the inputs live in their box: max over the box of S negative means
// no input sequence satisfies every state row -- a proof of infeasibility for ANY y >= 0 (Farkas), and the interior-point
// multipliers of the violated rows are the y that make S negative.  On the BASELINE planner draw (41 % infeasible QPs)
// the proof exists after 2.7 iterations on average; the divergence test (multipliers past 1e12) needed 9.3, up to 27.
//   max_box S = S(v) + sum over inputs a of (w_a > 0 ? w_a (hi_a - v_a) : w_a (-hi_a - v_a)),   w = J_s' y
// w comes from one adjoint sweep over the state-row multipliers alone (the structure of dual_infeasibility(): costate
// in registers, lanes >= NX receive B' lam); the stage gradients go through dZ, dead between the accept pass and the next
// forward sweep.  Entered only while the iterate still violates its rows after a step (a feasible problem: 0.14 times per
// solve on average).  Returns max_box S (wave-uniform).
template <int NOBS, int NMAX>
__device__ __forceinline__ double box_certificate(double* sm, const int* si, const Ctx& c, double delta_max, double a_max) {
    using L = Lay<NOBS, NMAX>;
    const int N = c.N, lane = c.lane;
    double S = 0.0;
    COORDS(e, ev, lane, N * L::NZ + L::NX) {
        const int k = e / L::NZ, a = e - k * L::NZ;
        const int rl = VROW(si)[L::vlo + e], rh = VROW(si)[L::vhi + e];
        const int il = rl >= 0 ? rl : 0, ih = rh >= 0 ? rh : 0;
        const double nl = LD(L::rnu + il), nh = LD(L::rnu + ih), cl = LD(L::rc + il), ch = LD(L::rc + ih);
        const bool st = ev && a < 6;                       // rows on states only: the input box is the domain
        const bool hl = st && rl >= 0, hh = st && rh >= 0;
        S += sel(hl, nl * cl, 0.0) + sel(hh, nh * ch, 0.0);
        LD(SINK(ev, L::dZ + e)) = sel(hh, nh, 0.0) - sel(hl, nl, 0.0);     // -J_s' y per stage coordinate
        (void)k;
    }
    SYNC();
    double mcol[L::NX];
#pragma unroll
    for (int i = 0; i < L::NX; i++) mcol[i] = LD(L::M + i * L::NZ + (lane < L::NZ ? lane : 0));
    const int la = lane < L::NZ ? lane : 0;
    const bool isu = lane >= L::NX && lane < L::NX + 2;
    const double hi = lane == L::NX ? delta_max : a_max;
    double tot = LD(L::dZ + N * L::NZ + la);
    double gk = LD(L::dZ + (N - 1) * L::NZ + la), uk = LD(L::Z + (N - 1) * L::NZ + la);
    for (int k = N - 1; k >= 0; k--) {
        const int kn = k >= 1 ? k - 1 : 0;
        const double gn = LD(L::dZ + kn * L::NZ + la), un = LD(L::Z + kn * L::NZ + la);
        double t = gk;
        if constexpr (ROWDPP<L>) {
            t = row_dot<L::NX, 0>(tot, mcol, t);
        } else {
#pragma unroll
            for (int i = 0; i < L::NX; i++) t += mcol[i] * lane_f64(tot, i);
        }
        tot = t;
        const double w = -tot;                            // input lanes: (J_s' y) of input a at stage k
        S += sel(isu, w * (sel(w > 0.0, hi, -hi) - uk), 0.0);
        gk = gn; uk = un;
    }
    SYNC();
    return wave_sum(S);
}

// Barrier-dependent assembly for the Newton system at barrier parameter mu:
//   per row   Sigma = nu/t,  w = nu - mu/t + Sigma*(c - t)          (one division per row)
//   per coord Hd = cost diag + Sigma of its bound rows + "current" CBF curvature,
//             hg = ga (reduced form) + J'w
//   per stage kS,kE = "next" CBF curvature  (- nu d hess g_{k+1}), added to P[4][4], P[5][5]
template <int NOBS, int NMAX>
__device__ __forceinline__ void assemble_newton(double* sm, const int* si, const Ctx& c, double mu) {
    using L = Lay<NOBS, NMAX>;
    const int N = c.N;
    ROWS(j, jv, c.lane, c.m) {
        (void)jv;
        const double t = LD(L::rt + j), nu = LD(L::rnu + j);
        const double rti = frcp(t);
        LD(L::rtt + j) = rti;                       // 1/t for the row-step pass (rtt is free until the line search)
        if constexpr (!L::SLIM) {
            const double sig = nu * rti;
            const bool on = LD(L::rsc + j) != 0.0;
            LD(L::rsig + j) = sel(on, sig, 0.0);
            const double cj = LD(L::rc + j);
            LD(L::rw + j) = sel(on, nu - mu * rti + sig * (cj - t), 0.0);
        } else {
            // slim [r4b]: w of the row for the coordinate pass below, parked in rdt (dead from the accept pass until the row steps that
            // follow the forward sweep; the backward sweep's P / T move in only after this function) -- the coordinate pass recomputed it per
            // coordinate and obstacle from four LDS reads.  Same expression as row_sig_w.
            const bool on = row_scale<L>(sm, si, j, N) != 0.0;
            const double sg = nu * rti, cj = LD(L::rc + j);
            LD(L::rdt + j) = sel(on, nu - mu * rti + sg * (cj - t), 0.0);
        }
    }
    SYNC();
    COORDS(e, ev, c.lane, N * L::NZ + L::NX) {   // stage N has states only (127 entries at N=12, 1 obstacle: two passes, not three)
        const int k = e / L::NZ, a = e - k * L::NZ;
        const bool isx = a < 6, isu = a >= L::NX && a < L::NX + 2, iss0 = a >= 6 && a < L::NX, iss1 = a >= L::NX + 2;
        const int o = iss0 ? a - 6 : (iss1 ? a - L::NX - 2 : 0);
        double g = LD(L::ga + e);
        const double cw = LD(L::cst + (isx ? a : (isu ? 6 + a - L::NX : 0)));
        double h = sel(isx || isu, 2.0 * cw, 0.0);
        // absent obstacle: pin its sigma_0 (state copy at k = 0) and sigma_{k+1} (input copy)
        h += ((iss0 && k == 0 && o >= c.nobs) || (iss1 && o >= c.nobs)) ? 1.0 : 0.0;
        int rl, rh;
        coord_rows<L>(si, c, e, k, a, rl, rh);
        const int il = rl >= 0 ? rl : 0, ih = rh >= 0 ? rh : 0;
        double sgl, sgh, wl, wh;
        if constexpr (L::SLIM) {   // a row found by coord_rows is present
            sgl = LD(L::rnu + il) * LD(L::rtt + il); wl = LD(L::rdt + il);
            sgh = LD(L::rnu + ih) * LD(L::rtt + ih); wh = LD(L::rdt + ih);
        } else {
            sgl = LD(L::rsig + il); sgh = LD(L::rsig + ih); wl = LD(L::rw + il); wh = LD(L::rw + ih);
        }
        h += sel(rl >= 0, sgl, 0.0) + sel(rh >= 0, sgh, 0.0);
        g += sel(rl >= 0, wl, 0.0) - sel(rh >= 0, wh, 0.0);
        if (NOBS) {
            const int kk = k < N ? k : N - 1;
#pragma unroll
            for (int ob = 0; ob < NOBS; ob++) {
                const int j = kk * L::NR + 8 + NOBS + ob;
                const double csj = cbf_scale<L>(sm, kk, ob);
                const double jca = LD(L::Jc + (kk * L::NO + ob) * L::NZ + a);
                double rsj_, rwj;
                if constexpr (L::SLIM) {
                    rwj = LD(L::rdt + j);
                } else {
                    row_sig_w<L>(sm, j, csj != 0.0, mu, LD(L::rt + j), LD(L::rnu + j), LD(L::rtt + j), LD(L::rc + j), rsj_, rwj);
                    (void)rsj_;
                }
                g += sel(k < N, jca * rwj, 0.0);
                const double cur = LD(L::rnu + j) * csj * c.om * LD(L::G + (kk * L::NO + ob) * 4 + (a == 4 ? 2 : 3));
                h += sel(k < N && (a == 4 || a == 5), cur, 0.0);
            }
        }
        const bool tail = k == N && a >= L::NX;
        LD(L::Hd + e) = sel(tail, 0.0, h);
        LD(SINK(ev, L::hg + e)) = sel(tail, 0.0, g);   // in place over ga: read-modify-write
    }
    if (NOBS) {
        const int k = c.lane < N ? c.lane : 0;          // N <= 24 < WAVE: one pass; lanes past N recompute stage 0
        double ks = 0.0, ke = 0.0;
#pragma unroll
        for (int o = 0; o < NOBS; o++) {
            const int j = k * L::NR + 8 + NOBS + o;
            const double nd = LD(L::rnu + j) * cbf_scale<L>(sm, k, o);
            ks -= nd * LD(L::G + (k * L::NO + o) * 4 + 0);
            ke -= nd * LD(L::G + (k * L::NO + o) * 4 + 1);
            if constexpr (L::sigS >= 0) LD(L::sigS + k * L::NO + o) = sel(LD(L::csc + k * L::NO + o) != 0.0, LD(L::rnu + j) * LD(L::rtt + j), 0.0);
        }
        LD(L::kS + 2 * k) = ks;
        LD(L::kE + 2 * k) = ke;
    }
    SYNC();
}

// Riccati backward sweep with regularisation dw on the input / sigma_0 diagonal.  Returns false if
// a pivot is not positive (wrong inertia).  On success Kk/kf hold the feedback and dZ[0] the step
// of the free initial components (sigma_0).
// [r6] REG = 1: the same sweep on the SECOND set of work arrays (Lay::P2 ..: the speculating wave of crx_solve_kernel<.., SPEC = 1>); CVX: the caller may
// ask for the convexified matrix (`convex`: the reverse-convex part kS / kE of the CBF curvature read as zero -- the one-wave kernel zeroes it in LDS
// instead, which two concurrent sweeps cannot).  <.., 0, false> is the code of rounds 1-5, operation for operation.
// KEEPF [r6]: the factor of every stage's Huu (unit-lower L, reciprocal pivots) is left behind in the stage's slice of Hd -- consumed by this stage's H
// phase, rewritten by the next assemble_newton -- for riccati_backward_vec, the second solve of the predictor-corrector iteration.
template <int NOBS, int NMAX, int UNR = 1, int REG = 0, bool CVX = false, bool KEEPF = false>
__device__ __forceinline__ bool riccati_backward(double* sm, const int* si, const Ctx& c, double dw, long long* tsub = nullptr, bool convex = false) {
    using L = Lay<NOBS, NMAX>;
    constexpr int NX = L::NX, NU = L::NU, NZ = L::NZ, HS = L::HS;
    constexpr int oP = REG ? L::P2 : L::P, opv = REG ? L::pv2 : L::pv, oT = REG ? L::T2 : L::T, oH = REG ? L::H2 : L::H, oKk = REG ? L::Kk2 : L::Kk,
                  okf = REG ? L::kf2 : L::kf, odZ = REG ? L::dZ2 : L::dZ;
    const int N = c.N, lane = c.lane;
    const long long qs = CLK();
    // [r4] The gradient p of the value function lives in REGISTERS, entry i in lane i: the update phase computes p_new there (its lane
    // map puts the gradient column first), and the H phase forms hv = M'p + hg with broadcast-FMAs (crx_wave.h row_dot) instead of NX
    // LDS reads per stage.  (pv in LDS is still written: the sigma_0 solve after the sweep reads it.)
    // [r6] PTR (NX <= 7: no or one obstacle): ALL of (P | p) lives in registers, one entry per lane -- lane 8 i + j holds P[i][j] (i, j < NX), lane
    // 8 NX + j holds p[j] -- and the T phase reads it with broadcast-FMAs inside the 8-lane halves of the DPP rows (crx_wave.h halfrow_dot6: lane
    // 8 i + c computes T[i][a(c)], its six operands P[i][0..5] sit in its own half-row).  The update phase computes P_new straight into that map (entry
    // (i, j) with i > j as its mirror image (j, i) does, operand for operand: symmetric to the bit as before), so the store -> load round trip of P
    // between the update and the T phase, one of the three of a stage, is gone; the sigma column of P goes from the update phase straight into T.
    // Same operations on the same operands in the same order as the LDS form: identical bits (tools/cbf_ab.py).  hv = M'p + hg is formed in the
    // DPP row that holds p (lanes 48 ..).  The NX + 1 feedback columns take the lanes the map leaves free (NX = 7: lanes 8 g + 7; NX = 6: 56 .. 62).
    constexpr bool PTR = CRX_PT_REG && ROWDPP<L> && L::UCNT == 1 && NX <= 7;
    constexpr bool PREG = !PTR && ROWDPP<L> && L::UCNT == 1;
    constexpr int PVF = (NX * 8) & 15;   // position of p[0] inside its DPP row (row 3 for NX = 6 and 7)
    static_assert(!PTR || (NX * 8) / 16 == 3, "p sits in DPP row 3, where hv is formed");
    // (the map is derived from an OPAQUE copy of the lane index, per call -- as the table look-up of the LDS form was: pure arithmetic on the lane index
    // would be hoisted out of the interior-point loop into ~10 registers the one-obstacle instantiation does not have: 36 B of scratch per lane)
    int lo = lane;
    if constexpr (PTR) asm volatile("" : "+v"(lo));
    const int g8 = lo >> 3, c8 = lo & 7;
    const bool isPm = g8 < NX && c8 < NX, isPv = g8 == NX && c8 < NX;
    double preg = 0.0, pn = 0.0;
    // terminal: P_N = diag(Hd[N][0..NX)) + stage N-1 extras on (s_N, ey_N);  p_N = hg[N]
    {
        double kSn = NOBS ? LD(L::kS + 2 * (N - 1)) : 0.0, kEn = NOBS ? LD(L::kE + 2 * (N - 1)) : 0.0;
        if constexpr (CVX) { kSn = sel(convex, 0.0, kSn); kEn = sel(convex, 0.0, kEn); }
        kEn += 2.0 * LD(L::wc + N - 1);
        if constexpr (PTR) {
            const double hd = LD(L::Hd + N * NZ + seli(isPm, g8, 0));
            const double hgN = LD(L::hg + N * NZ + seli(c8 < NX, c8, 0));
            pn = sel(isPv, hgN, sel(isPm && g8 == c8, hd + sel(g8 == 4, kSn, sel(g8 == 5, kEn, 0.0)), 0.0));
            if (NOBS) {   // sigma columns of T: sigma_k -> 0 (constant over the sweep), sigma_{k+1} -> P[:, 6]
                LD(SINK(lo < NX, oT + lo * L::TS + 6)) = 0.0;
                LD(SINK(isPm && c8 == 6, oT + g8 * L::TS + NX + 2)) = pn;
            }
        } else {
#pragma unroll
            for (int q_ = 0; q_ < (NX * NX + WAVE - 1) / WAVE; q_++) {
                const int e0 = lane + q_ * WAVE;
                const int e = e0 < NX * NX ? e0 : 0;            // lanes past the matrix recompute entry 0
                const int i = e / NX, j = e - i * NX;
                const double hd = LD(L::Hd + N * NZ + i);
                LD(oP + e) = sel(i == j, hd + sel(i == 4, kSn, sel(i == 5, kEn, 0.0)), 0.0);
            }
            const double hgN = LD(L::hg + N * NZ + (lane < NX ? lane : 0));
            LD(SINK(lane < NX, opv + lane)) = hgN;
            preg = hgN;
        }
    }
    SYNC();
    bool ok = true;
    // The sweep is straight-line code: every lane-dependent decision (which entry a lane owns, where it goes) is
    // decoded ONCE here into addresses; lanes without an entry store to the sink L::dmy instead of being masked off.
    // A predicated region costs ~30 cycles of exec-mask traffic, and -- worse -- the compiler sinks the LDS loads of
    // values only such a region uses into it, which turned the update phase into three serialized LDS round trips.
    constexpr bool FULL = NZ * NZ <= WAVE;
    constexpr int NTRI = FULL ? NZ * NZ : NZ * (NZ + 1) / 2;
    constexpr int HCNT = (NTRI + WAVE - 1) / WAVE;
    int hr[HCNT], ha[HCNT], hst[HCNT], hst2[HCNT];
#pragma unroll
    for (int q = 0; q < HCNT; q++) {
        const int e0 = lane + q * WAVE;
        const int e = e0 < NTRI ? e0 : 0;
        if (FULL) { hr[q] = e / NZ; ha[q] = e - hr[q] * NZ; }
        else if constexpr (L::SLIM) tri_decode(e, hr[q], ha[q]);
        else { const int pk = si[L::triH + e]; hr[q] = pk >> 8; ha[q] = pk & 255; }
        hst[q] = SINK(e0 < NTRI, oH + hr[q] * HS + ha[q]);
        hst2[q] = SINK(CRX_H_MIRROR && !FULL && e0 < NTRI, oH + ha[q] * HS + hr[q]);
    }
    const bool hvl = PTR ? (lane >= 48 && lane < 48 + NZ) : lane < NZ;   // the lanes that form hv (PTR: in the DPP row of p)
    const int lz = hvl ? (PTR ? lane - 48 : lane) : 0;
    const int hvst = SINK(hvl, oH + lz * HS + NZ);
    // the entries of the model matrix each lane multiplies with in the T and H phases, in registers
    constexpr int TCNT = (NX * 8 + WAVE - 1) / WAVE;   // 1 for NX <= 8, 2 for NX = 9
    double mT[TCNT][6], mH[HCNT][NX];
    int tld[TCNT], tst[TCNT];
#pragma unroll
    for (int q = 0; q < TCNT; q++) {
        const int e = lane + q * WAVE;
        const int cc = e & 7;
        const int a = cc < 6 ? cc : NX + (cc - 6);
        const int i = (e >> 3) < NX ? (e >> 3) : 0;
        tld[q] = oP + i * NX;
        tst[q] = SINK(e < NX * 8, oT + i * L::TS + a);
#pragma unroll
        for (int j = 0; j < 6; j++) mT[q][j] = LD(L::M + j * NZ + a);
    }
#pragma unroll
    for (int q = 0; q < HCNT; q++)
#pragma unroll
        for (int i = 0; i < NX; i++) mH[q][i] = LD(L::M + i * NZ + hr[q]);
    double m5a[HCNT], m5r[HCNT];   // planner: row 5 of the model matrix at the entry's column / row (coupling cost), stage-invariant too [r4]
#pragma unroll
    for (int q = 0; q < HCNT; q++) { m5a[q] = NOBS ? 0.0 : LD(L::M + 5 * NZ + ha[q]); m5r[q] = NOBS ? 0.0 : LD(L::M + 5 * NZ + hr[q]); }
    double mz[NX];   // column lz of the model matrix (the gradient column hv = M'p + hg): stage-invariant like mT / mH [r4]
#pragma unroll
    for (int i = 0; i < NX; i++) mz[i] = LD(L::M + i * NZ + lz);
    // sigma columns of T (NOBS > 0): sigma_k -> 0, sigma_{k+1} -> P[:, 6+o]
    constexpr int T2CNT = NOBS ? (NX * 2 * NOBS + WAVE - 1) / WAVE : 1;   // one pass up to three obstacles, three for six [r4]
    int t2ld[T2CNT], t2st[T2CNT];
    bool t2nxt[T2CNT];
#pragma unroll
    for (int q = 0; q < T2CNT; q++) {
        t2ld[q] = oP; t2st[q] = L::dmy; t2nxt[q] = false;
        if (NOBS) {
            const int l2 = lane + q * WAVE;
            const bool v2 = l2 < NX * 2 * NOBS;
            const int e2 = v2 ? l2 : 0;
            const int i2 = e2 / (2 * L::NO), cc = e2 - i2 * (2 * L::NO);
            t2nxt[q] = cc >= NOBS;
            const int o = t2nxt[q] ? cc - NOBS : cc;
            t2ld[q] = oP + i2 * NX + 6 + o;
            t2st[q] = SINK(v2, oT + i2 * L::TS + seli(t2nxt[q], NX + 2 + o, 6 + o));
        }
    }
    // update phase: lane map [0, NP) the upper triangle of P_new incl. the gradient column (i <= j <= NX),
    // [NP, NP+NX+1) one feedback column each (<= 64 lanes for NX <= 9: one pass; the six-obstacle instantiation takes two [r4])
    constexpr int NP = NX * (NX + 1) / 2 + NX, UCNT = L::UCNT;
    int yiA[UCNT], yjA[UCNT], s0A[UCNT], pst1[UCNT], pst2[UCNT], kstr[UCNT], kstep[UCNT], kst[UCNT];
    bool exSl[UCNT], exEl[UCNT];
    // [r6] 0 / 1 lane masks: t + x on the lanes of a set, t on the others, is fma(mask, x, t) -- one rounding of the same sum (x finite: 0 * x = 0) --
    // instead of two v_cndmask_b32 and an add per use and stage
    double mSl[UCNT], mEl[UCNT], mdg[HCNT];
#pragma unroll
    for (int q = 0; q < HCNT; q++) mdg[q] = hr[q] == ha[q] ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < UCNT; q++) {
        const int l = lane + q * WAVE;
        int ui, uj; bool isP, isK;                   // feedback lanes: column uj, ui = 0 (unused)
        if constexpr (PTR) {
            isP = isPm || isPv;
            isK = NX == 7 ? c8 == 7 : (g8 == 7 && c8 <= NX);
            ui = seli(isPm, g8 < c8 ? g8 : c8, seli(isPv, c8, 0));
            uj = seli(isPm, g8 < c8 ? c8 : g8, seli(isPv, NX, seli(isK, NX == 7 ? g8 : c8, 0)));
        } else {
            const int upk = UPDP(si, l);
            ui = upk >> 8; uj = upk & 255;
            isP = l < NP; isK = !isP && l < NP + NX + 1;
        }
        const bool gcol = uj >= NX;                   // gradient column = column NZ of H
        const int ujj = gcol ? NZ : uj;
        // [r6] H(ui, uj) is read from the LOWER triangle (uj >= ui), like every other operand of the phase: the H phase need not mirror its entries
        // (one LDS store per pass and stage).  The planner's H is a full matrix, every entry its own sum: its (ui, uj) stays where it was.
        yiA[q] = oH + NX * HS + ui; yjA[q] = oH + NX * HS + ujj;
        s0A[q] = (FULL || CRX_H_MIRROR) ? oH + ui * HS + ujj : seli(gcol, oH + ui * HS + ujj, oH + uj * HS + ui);
        pst1[q] = SINK(isP, seli(gcol, opv + ui, oP + ui * NX + uj));
        pst2[q] = SINK(isP, seli(gcol, opv + ui, oP + uj * NX + ui));
        kstr[q] = seli(isK, seli(gcol, 1, NX), 0); kstep[q] = seli(isK, seli(gcol, NU, NU * NX), 0);
        kst[q] = SINK(isK, seli(gcol, okf, oKk + uj) + (N - 1) * kstep[q]);
        exSl[q] = isP && !gcol && ui == uj && ui == 4; exEl[q] = isP && !gcol && ui == uj && ui == 5;
        mSl[q] = exSl[q] ? 1.0 : 0.0; mEl[q] = exEl[q] ? 1.0 : 0.0;
    }
    if (tsub) tsub[2] += CLK() - qs;   // set-up of the sweep (terminal P, lane maps, stage-invariant operands)
#pragma unroll UNR
    for (int k = N - 1; k >= 0; k--) {
        long long q0 = CLK();
        // (Forming H = M'PM in ONE phase straight from P -- NX^2 FMAs per entry, no T -- was tried twice, before and after
        // the sweep became straight-line code: 16 % / 6 % SLOWER on cfg2 / cfg3; 49 broadcast loads and FMAs per lane cost
        // more than the LDS round trip they save, and the registers cost a resident wave.)
        // T = P M, one pass: only the x- and u-columns of M carry numbers (6+2 columns, NX*8 <= 64
        // dot products of length 6); the sigma_k columns of T are zero and the sigma_{k+1} columns
        // are copies of P's sigma columns (M = [A 0 B 0; 0 0 0 I]).
        if constexpr (PTR) {
            __builtin_amdgcn_wave_barrier();   // scheduling region boundary only (the update phase in front needs no exchange with this one)
            LD(tst[0]) = halfrow_dot6(pn, mT[0], (lane & 8) != 0);
        } else {
            double pl[TCNT][6];
#pragma unroll
            for (int q = 0; q < TCNT; q++)
#pragma unroll
                for (int j = 0; j < 6; j++) pl[q][j] = LD(tld[q] + j);
            double p2[T2CNT];
#pragma unroll
            for (int q = 0; q < T2CNT; q++) p2[q] = NOBS ? LD(t2ld[q]) : 0.0;
            LOADS_DONE();
#pragma unroll
            for (int q = 0; q < TCNT; q++) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < 6; j++) t += pl[q][j] * mT[q][j];
                LD(tst[q]) = t;
            }
            if (NOBS) {
#pragma unroll
                for (int q = 0; q < T2CNT; q++) LD(t2st[q]) = sel(t2nxt[q], p2[q], 0.0);
            }
        }
        SYNC();
        long long q1 = CLK();
        // H = M'T + stage terms, lower triangle only (NZ(NZ+1)/2 <= 105 entries), mirrored on store;
        // hv = M'p + hg, stored as column NZ of H.  NZ*NZ <= 64 (no obstacle): one lane per entry of the
        // full matrix, no mirroring needed.
        {
            // every operand of the phase is loaded first (LOADS_DONE pins that order: left alone, the scheduler trickles
            // the loads out between the FMAs and the two chains pay the LDS latency one after the other)
            // (PTR: the gradient of the stage FIRST -- hv = M'p + hg needs nothing else from LDS, its broadcast-FMAs run while the other loads are in flight)
            double hvs = LD(L::hg + k * NZ + lz), pvv[NX];
            if constexpr (PTR) __builtin_amdgcn_sched_barrier(0);        // (left alone the scheduler issues this load last)
            const double kc = (NOBS == 0) ? 2.0 * LD(L::wc + k) : 0.0;   // coupling cost exists in planner mode only
            double tc[HCNT][NX], hd[HCNT], jr[HCNT][L::NO], ja[HCNT][L::NO], rs[L::NO];
#pragma unroll
            for (int q = 0; q < HCNT; q++) {
#pragma unroll
                for (int i = 0; i < NX; i++) tc[q][i] = LD(oT + i * L::TS + ha[q]);
                hd[q] = LD(L::Hd + k * NZ + hr[q]);
                if (NOBS) {
#pragma unroll
                    for (int o = 0; o < NOBS; o++) {
                        jr[q][o] = LD(L::Jc + (k * L::NO + o) * NZ + hr[q]);
                        ja[q][o] = LD(L::Jc + (k * L::NO + o) * NZ + ha[q]);
                    }
                }
            }
            if (NOBS) {
#pragma unroll
                for (int o = 0; o < NOBS; o++) {
                    if constexpr (L::sigS >= 0) {
                        rs[o] = LD(L::sigS + k * L::NO + o);
                    } else if constexpr (L::SLIM) {   // Sigma of the CBF row (k, o): nu / t, 0 for an absent row
                        const int j = k * L::NR + 8 + NOBS + o;
                        rs[o] = sel(LD(L::csc + k * L::NO + o) != 0.0, LD(L::rnu + j) * LD(L::rtt + j), 0.0);
                    } else {
                        rs[o] = LD(L::rsig + k * L::NR + 8 + NOBS + o);
                    }
                }
            }
            if constexpr (!PREG && !PTR) {
#pragma unroll
                for (int i = 0; i < NX; i++) pvv[i] = LD(opv + i);
            }
            LOADS_DONE();
            if constexpr (PTR) {
                hvs = row_dot<NX, PVF>(pn, mz, hvs);
                __builtin_amdgcn_sched_barrier(0);
            }
            double hs[HCNT];
#pragma unroll
            for (int q = 0; q < HCNT; q++) {
                const int r = hr[q], a = ha[q];
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < NX; i++) t += mH[q][i] * tc[q][i];
                const double dg = hd[q] + sel(r >= NX || (k == 0 && r >= 6), dw, 0.0);
                if constexpr (CRX_MASK_FMA) t = fma(mdg[q], dg, t);
                else t += sel(r == a, dg, 0.0);
                if (NOBS) {
#pragma unroll
                    for (int o = 0; o < NOBS; o++) t += rs[o] * jr[q][o] * ja[q][o];
                } else {
                    // kC (m_e - e_ey)(m_e - e_ey)' with the m_e m_e' part already inside P[5][5]
                    t -= kc * (sel(r == 5, m5a[q], 0.0) + sel(a == 5, m5r[q], 0.0) - ((r == 5 && a == 5) ? 1.0 : 0.0));
                }
                hs[q] = t;
            }
            if constexpr (PTR) {
            } else if constexpr (PREG) {
                hvs = row_dot<NX, 0>(preg, mz, hvs);
            } else {
#pragma unroll
                for (int i = 0; i < NX; i++) hvs += mz[i] * pvv[i];
            }
#pragma unroll
            for (int q = 0; q < HCNT; q++) {
                LD(hst[q]) = hs[q];
                if (!FULL && CRX_H_MIRROR) LD(hst2[q]) = hs[q];
            }
            LD(hvst) = hvs;
        }
        SYNC();
        long long q2 = CLK();
        // Huu = L D L' (every lane factorises it itself: NU <= 5, broadcast LDS reads; unit-lower L, reciprocal
        // pivots) and the factorised update (block-Cholesky form) in ONE LDS round trip: Y = L^{-1} Hux,
        // P_new = Hxx - Y' D^{-1} Y, K = -L^{-T} D^{-1} Y; column NX of the lane map is the gradient column
        // (p_new, kff).  P_new is formed from the SAME factor for (i,j) and (j,i): symmetric positive semi-definite
        // by construction.  P/pv are not read in this phase (T and hv are done), so they are overwritten in place.
        {
            double Lf[NU][NU], Dp[NU], rD[NU], yi[UCNT][NU], yj[UCNT][NU], t[UCNT];
#pragma unroll
            for (int a = 0; a < NU; a++)
#pragma unroll
                for (int b2 = 0; b2 <= a; b2++) Lf[a][b2] = LD(oH + (NX + a) * HS + NX + b2);
            // (Huu FIRST: the pivot chain -- the longest of the phase -- starts when these are back, while the other loads are in flight; left alone the
            // scheduler issues them fourth)
            if constexpr (CRX_HUU_FIRST) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < UCNT; q++) {
#pragma unroll
                for (int a = 0; a < NU; a++) { yi[q][a] = LD(yiA[q] + a * HS); yj[q][a] = LD(yjA[q] + a * HS); }
                t[q] = LD(s0A[q]);
            }
            const int km = k >= 1 ? k - 1 : 0;     // stage k-1 extras on (s_k, ey_k), wave-uniform
            // (the coupling weight wc is a planner quantity: identically zero with obstacles, and not loaded there)
            double kSv = NOBS ? LD(L::kS + 2 * km) : 0.0, kEv = NOBS ? LD(L::kE + 2 * km) : 0.0;
            const double wcv = NOBS ? 0.0 : LD(L::wc + km);
            if constexpr (CVX) { kSv = sel(convex, 0.0, kSv); kEv = sel(convex, 0.0, kEv); }
            LOADS_DONE();
            const double exS = sel(k >= 1, kSv, 0.0), exE = sel(k >= 1, NOBS ? kEv : kEv + 2.0 * wcv, 0.0);
            // with C = L D kept beside L (C[i][q] = L[i][q] D[q]) every update on the pivot chain is ONE fma:
            // d_j = H_jj - sum_q L[j][q] C[j][q],  C[i][j] = H_ij - sum_q L[i][q] C[j][q],  L[i][j] = C[i][j] / d_j
            double Cf[NU][NU];
#pragma unroll
            for (int j = 0; j < NU; j++) {
                double d = Lf[j][j];
#pragma unroll
                for (int q = 0; q < j; q++) d = fma(-Lf[j][q], Cf[j][q], d);
                if (!(d > 0.0)) ok = false;
                Dp[j] = d;
                rD[j] = frcp(d);
#pragma unroll
                for (int i = j + 1; i < NU; i++) {
                    double u = Lf[i][j];
#pragma unroll
                    for (int q = 0; q < j; q++) u = fma(-Lf[i][q], Cf[j][q], u);
                    Cf[i][j] = u;
                    Lf[i][j] = u * rD[j];
                }
            }
            if constexpr (KEEPF) {
                static_assert(NU * (NU + 1) / 2 <= NZ, "the factor fits the stage's slice of Hd");
                int f_ = 0;
#pragma unroll
                for (int a = 0; a < NU; a++) {
                    LD(SINK(lane == 0, L::Hd + k * NZ + f_)) = rD[a]; f_++;
#pragma unroll
                    for (int b2 = 0; b2 < a; b2++) { LD(SINK(lane == 0, L::Hd + k * NZ + f_)) = Lf[a][b2]; f_++; }
                }
            }
#pragma unroll
            for (int q = 0; q < UCNT; q++) {
#pragma unroll
                for (int a = 1; a < NU; a++) {
#pragma unroll
                    for (int qq = 0; qq < a; qq++) { yi[q][a] -= Lf[a][qq] * yi[q][qq]; yj[q][a] -= Lf[a][qq] * yj[q][qq]; }
                }
#pragma unroll
                // (yj carries MINUS D^{-1} y from here on: the feedback -L^{-T} D^{-1} y is then stored as it comes out of the back substitution --
                // negation commutes with every rounding below, the bits are those of the negate-at-the-store form [r4])
                for (int a = 0; a < NU; a++) { yj[q][a] *= -rD[a]; t[q] = fma(yi[q][a], yj[q][a], t[q]); }
                if constexpr (CRX_MASK_FMA) t[q] = fma(mSl[q], exS, fma(mEl[q], exE, t[q]));   // (at most one of the two masks is set)
                else t[q] += sel(exSl[q], exS, sel(exEl[q], exE, 0.0));
                if constexpr (PTR) {
                    pn = t[0];                     // (P_new | p_new) stays in registers; its sigma column is the next stage's sigma_{k+1} column of T
                    if (NOBS) LD(SINK(isPm && c8 == 6, oT + g8 * L::TS + NX + 2)) = pn;
                } else {
                    LD(pst1[q]) = t[q];
                    LD(pst2[q]) = t[q];
                }
                if constexpr (PREG) preg = t[0];   // lanes 0 .. NX-1: p_new (gradient column of the lane map)
#pragma unroll
                for (int a = NU - 2; a >= 0; a--) {
#pragma unroll
                    // (an explicit fma: with P_new in registers the last stage of a problem without obstacles has no use for t, the scaled yj
                    // above is left with ONE use, and -ffp-contract would fold its multiplication into this line instead -- fma(y, -1/D, -(L yj)) -- another
                    // rounding of the stage-0 feedback than in every other stage and than in the LDS form: found as last-bit differences in 46 % of
                    // the planner QPs, profiles/r06_ptreg.txt)
                    for (int qq = a + 1; qq < NU; qq++) yj[q][a] = fma(-Lf[qq][a], yj[q][qq], yj[q][a]);
                }
#pragma unroll
                for (int a = 0; a < NU; a++) LD(kst[q] + a * kstr[q]) = yj[q][a];
                kst[q] -= kstep[q];
            }
        }
        if (!ok) break;  // uniform: every lane computed the same pivots (the entries just stored are discarded with the sweep)
        SYNC();
        long long q4 = CLK();
        if (tsub) { tsub[0] += q1 - q0; tsub[1] += q2 - q1; tsub[3] += q4 - q2; }
    }
    if (!ok) { SYNC(); return false; }
    if constexpr (PTR && NOBS > 0) {   // (P_0 | p_0) of the sigma_0 solve below: the only reader of P and pv outside the sweep
        LD(SINK(isPm, oP + g8 * NX + c8)) = pn;
        LD(SINK(isPv, opv + c8)) = pn;
        SYNC();
    }
    // free initial components sigma_0: minimise 1/2 d'P d + p'd over them (x_0 is fixed)
    if (lane < NZ) LD(odZ + lane) = 0.0;
    if (NOBS) {
        double Ls[L::NO][L::NO], y[L::NO], Ds[L::NO];
#pragma unroll
        for (int a = 0; a < NOBS; a++) {
#pragma unroll
            for (int b = 0; b <= a; b++) Ls[a][b] = LD(oP + (6 + a) * NX + 6 + b);
            y[a] = -LD(opv + 6 + a);
        }
#pragma unroll
        for (int j = 0; j < NOBS; j++) {
            double d = Ls[j][j];
#pragma unroll
            for (int q = 0; q < j; q++) d -= Ls[j][q] * Ls[j][q] * Ds[q];
            if (!(d > 0.0)) ok = false;
            Ds[j] = d;
#pragma unroll
            for (int i = j + 1; i < NOBS; i++) {
                double t = Ls[i][j];
#pragma unroll
                for (int q = 0; q < j; q++) t -= Ls[i][q] * Ls[j][q] * Ds[q];
                Ls[i][j] = t / d;
            }
        }
        if (ok) {
#pragma unroll
            for (int a = 1; a < NOBS; a++) {
#pragma unroll
                for (int q = 0; q < a; q++) y[a] -= Ls[a][q] * y[q];
            }
#pragma unroll
            for (int a = 0; a < NOBS; a++) y[a] /= Ds[a];
#pragma unroll
            for (int a = NOBS - 2; a >= 0; a--) {
#pragma unroll
                for (int q = a + 1; q < NOBS; q++) y[a] -= Ls[q][a] * y[q];
            }
            SYNC();
            if (lane == 0) {
#pragma unroll
                for (int a = 0; a < NOBS; a++) LD(odZ + 6 + a) = y[a];
            }
        }
    }
    SYNC();
    return ok;
}

// [r6] Second solve with the factorisation riccati_backward<.., KEEPF> left behind: the VECTOR recursion alone, for another Newton gradient hg
// (the corrector of the predictor-corrector iteration).  Per stage, lanes 0 .. NZ-1:  hv = M'p + hg_k  (one broadcast-FMA dot product, p in lanes
// 0 .. NX-1),  kff = -Huu^{-1} hu  from the stored factor (L, 1 / D: Hd[k][0 .. 2]),  p <- hx + K'hu  with the stored feedback K (= -Huu^{-1} Hux).
// ~30 instructions per stage against ~135 of the full stage.  Writes kf; the matrices P are not needed again.
template <int NOBS, int NMAX, int UNR = 1>
__device__ __forceinline__ void riccati_backward_vec(double* sm, const Ctx& c) {
    using L = Lay<NOBS, NMAX>;
    static_assert(NOBS == 0 && L::NU == 2, "the second solve exists for the all-linear problems (two inputs per stage)");
    constexpr int NX = L::NX, NU = L::NU, NZ = L::NZ;
    const int N = c.N, lane = c.lane;
    if (!CRX_SWEEP_MASK || lane < NZ) {
        const bool in = CRX_SWEEP_MASK || lane < NZ;
        const int la = in ? lane : 0, lx = la < NX ? la : 0;
        double mcol[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) mcol[i] = LD(L::M + i * NZ + la);
        double p = LD(L::hg + N * NZ + la);             // lanes < NX: p_N = the terminal gradient
        double gk = LD(L::hg + (N - 1) * NZ + la);
        auto stage = [&](int k) {
            const double gn = LD(L::hg + (k >= 1 ? k - 1 : 0) * NZ + la);
            const double rD0 = LD(L::Hd + k * NZ + 0), rD1 = LD(L::Hd + k * NZ + 1), L10 = LD(L::Hd + k * NZ + 2);
            const double k0 = LD(L::Kk + (k * NU + 0) * NX + lx), k1 = LD(L::Kk + (k * NU + 1) * NX + lx);
            double hv = gk;
            if constexpr (ROWDPP<L>) {
                hv = row_dot<NX, 0>(p, mcol, hv);
            } else {
#pragma unroll
                for (int i = 0; i < NX; i++) hv += mcol[i] * lane_f64(p, i);
            }
            const double hu0 = lane_f64(hv, NX), hu1 = lane_f64(hv, NX + 1);
            const double y1 = hu1 - L10 * hu0;
            const double z1 = y1 * rD1;
            const double z0 = hu0 * rD0 - L10 * z1;
            LD(SINK(in && lane < NU, L::kf + k * NU + lane)) = sel(lane == 0, -z0, -z1);
            p = hv + k0 * hu0 + k1 * hu1;               // lanes < NX: p_k (the other lanes carry numbers nobody reads)
            gk = gn;
            STAGE_FENCE();
        };
        if constexpr (UNR > 1) {
#pragma unroll UNR
            for (int k = N - 1; k >= 0; k--) stage(k);
        } else {
            for (int k = N - 1; k >= 0; k--) stage(k);
        }
        LD(SINK(in, L::dZ + lane)) = 0.0;               // dx_0 = 0 (x_0 is fixed; no free sigma_0 without obstacles)
    }
    SYNC();
}

// forward sweep: du = K dx + kff, dx_{k+1} = M [dx; du]; fills dZ for every stage.  The recursion lives in registers:
// lane j < NX carries dx_k[j], lane NX + a computes du_a; both are broadcast with v_readlane, the gains of the next
// stage are loaded while this one computes, LDS only receives the result (it was two dependent LDS round trips per
// stage).  Same operations in the same order as the LDS version: identical bits.  (A one-broadcast-per-stage variant --
// lane i holding row i of the closed-loop map [M_x + M_u K_k | M_u kff_k], formed from K_{k+1} while stage k runs -- has
// the shorter chain but costs NU (NX+1) more loads and FMAs per lane and stage: measured 6-13 % SLOWER.)
template <int NOBS, int NMAX, int UNR = 1>
__device__ __forceinline__ void riccati_forward(double* sm, const Ctx& c) {
    using L = Lay<NOBS, NMAX>;
    constexpr int NX = L::NX, NU = L::NU, NZ = L::NZ;
    const int N = c.N;
    int lane = c.lane;
#if CRX_SWEEP_LOCAL_LANE & 1
    // the handful of lane-derived addresses and masks of this sweep are formed HERE, every call: left to the optimiser they are hoisted out of
    // the interior-point loop and, at the register limit of two waves per SIMD, parked in scratch (one reload each per iteration) [r4]
    if (NOBS > 0) asm volatile("" : "+v"(lane));
#endif
    if constexpr (ROWDPP<L> && CRX_SWEEP_MASK && CRX_FWD_ONE_DOT && NOBS > 0) {   // (planner instantiations: measured 1.5 % slower with it, cfg3)
        // [r4] ONE dot product per lane and stage.  A state lane i < NX needs x_{k+1}[i] = M[i][0..NX) x_k + M[i][NX..) du_k, an input lane
        // NX + a needs du_k[a] = kff_k[a] + K_k[a] x_k: both are `c0 + coef . x_k` with lane-specific (c0, coef) -- the coefficients come
        // from a lane-specific ADDRESS (row i of M, every stage; row a of K_k), not from a select, so the sweep no longer runs both
        // chains in every lane (17 FMAs -> 10).  The input term M[i][NX..) du_k follows with zero coefficients in the input lanes, whose
        // value is thereby left as it is.  Same terms in the same order as the two-chain form: identical bits (signs of zero aside).
        if (lane < NZ) {
            const bool isx = lane < NX;
            const int ua = isx ? 0 : lane - NX;
            int cad = isx ? L::M + lane * NZ : L::Kk + ua * NX;            // + k NU NX in the input lanes
            const int cstep = isx ? 0 : NU * NX;
            const int kfad = L::kf + ua;                                     // state lanes read kff_k[0] and multiply it with zero
            const int sad = L::dZ + lane + (isx ? NZ : 0);                   // state lanes store x_{k+1}, input lanes du_k
            const double one = isx ? 0.0 : 1.0;
            double mu_[NU];
#pragma unroll
            for (int a = 0; a < NU; a++) mu_[a] = sel(isx, LD(L::M + (isx ? lane : 0) * NZ + NX + a), 0.0);
            double y = LD(L::dZ + (isx ? lane : 0));   // dx_0: zero but for the free sigma_0 (riccati_backward)
            double coef[NX], kfv;                      // this stage's coefficients: loaded while the previous stage computes
#pragma unroll
            for (int j = 0; j < NX; j++) coef[j] = LD(cad + j);
            kfv = LD(kfad);
            auto stage = [&](int k) {
                double cnx[NX], kfn;
                const int kn = k + 1 < N ? k + 1 : k;
                cad += (k + 1 < N) ? cstep : 0;
#pragma unroll
                for (int j = 0; j < NX; j++) cnx[j] = LD(cad + j);
                kfn = LD(kfad + kn * NU);
                double acc = fma(kfv, one, 0.0);
                acc = row_dot<NX, 0>(y, coef, acc);
                const double du = acc;                                       // lanes NX ..: du_k
                acc = row_dot<NU, NX>(du, mu_, acc);
                y = acc;
                LD(sad + k * NZ) = y;
#pragma unroll
                for (int j = 0; j < NX; j++) coef[j] = cnx[j];
                kfv = kfn;
                STAGE_FENCE();
            };
            if constexpr (UNR > 1) {
#pragma unroll UNR
                for (int k = 0; k < N; k++) stage(k);
            } else {
                for (int k = 0; k < N; k++) stage(k);
            }
            LD(SINK(!isx, L::dZ + N * NZ + lane)) = 0.0;                     // no inputs at stage N
        }
        SYNC();
        return;
    }
    if (!CRX_SWEEP_MASK || lane < NZ) {   // one predicated region, lanes 0 .. NZ-1: see dual_infeasibility [r4]
    constexpr bool MK = CRX_SWEEP_MASK;
    const bool in = MK || lane < NZ;
    double mrow[NZ];      // row `lane` of the model matrix, in registers across the sweep
#pragma unroll
    for (int j = 0; j < NZ; j++) mrow[j] = LD(L::M + (lane < NX ? lane : 0) * NZ + j);
    const bool isu = lane >= NX && in;
    const int ua = isu ? lane - NX : 0;
    double zx = LD(L::dZ + (lane < NX ? lane : 0));   // dx_0: zero but for the free sigma_0 (riccati_backward)
    double kr[NX], kfa;
#pragma unroll
    for (int j = 0; j < NX; j++) kr[j] = LD(L::Kk + ua * NX + j);
    kfa = LD(L::kf + ua);
    auto stage = [&](int k) {
        double krn[NX], kfn;                           // next stage's feedback row: off the dependent chain
        const int kn = k + 1 < N ? k + 1 : k;
#pragma unroll
        for (int j = 0; j < NX; j++) krn[j] = LD(L::Kk + (kn * NU + ua) * NX + j);
        kfn = LD(L::kf + kn * NU + ua);
        double du = kfa, xn = 0.0;
        if constexpr (ROWDPP<L>) {   // broadcast and FMA in one instruction (crx_wave.h row_dot): same terms in the same order
            du = row_dot<NX, 0>(zx, kr, du);
            xn = row_dot<NX, 0>(zx, mrow, xn);
            xn = row_dot<NU, NX>(du, mrow + NX, xn);
        } else {
            double xs[NX];
#pragma unroll
            for (int j = 0; j < NX; j++) xs[j] = lane_f64(zx, j);
#pragma unroll
            for (int j = 0; j < NX; j++) du += kr[j] * xs[j];
#pragma unroll
            for (int j = 0; j < NX; j++) xn += mrow[j] * xs[j];
#pragma unroll
            for (int a = 0; a < NU; a++) xn += mrow[NX + a] * lane_f64(du, NX + a);
        }
        LD(SINK(in, L::dZ + k * NZ + lane)) = sel(isu, du, zx);
        zx = xn;
#pragma unroll
        for (int j = 0; j < NX; j++) kr[j] = krn[j];
        kfa = kfn;
        STAGE_FENCE();
    };
    if constexpr (UNR > 1) {   // fixed horizon [r4]: straight-line code -- the hand-over kr <- krn is a renaming instead of NX + 1 register copies
#pragma unroll UNR             // per stage, the stage addresses are immediates
        for (int k = 0; k < N; k++) stage(k);
    } else {
        for (int k = 0; k < N; k++) stage(k);
    }
    LD(SINK(in, L::dZ + N * NZ + lane)) = sel(lane < NX, zx, 0.0);
    }
    SYNC();
}

// Restoration (CBF instantiations only), entered when the filter line search finds no acceptable step -- where IPOPT
// switches to its restoration phase.  What jams on crash states (ego inside, or about to enter, an obstacle's unsafe
// set) is the collapse of the slacks t_j of CBF rows that stay violated.  Every CBF row reads
//     G_i(x_i, x_{i+1}) + (1 - alpha) sigma_i - sigma_{i+1} >= 0        (control.py:544-558)
// with free sigma >= 0, so at the current inputs the least-violation point exists in closed form with zero violation:
// raise the slacks stage by stage from the end of the horizon, sigma_i >= (sigma_{i+1} - G_i + push_i) / (1 - alpha).
// Same arithmetic, in the same order, as oracle/crx_oracle.c restore_slacks().  Runs once or twice in the life of a
// crash problem and never otherwise.  (Not a real call: a non-inlined callee makes the kernel reserve the full register file, one wave per SIMD.)
// G_i goes through the (then dead) row-step array rdt; the recursion itself is serial and runs on every lane alike.
template <int NOBS, int NMAX>
__device__ __forceinline__ bool restore_slacks(double* sm, const Ctx& c, double slack_push) {
    using L = Lay<NOBS, NMAX>;
    constexpr int NX = L::NX, NZ = L::NZ, NR = L::NR;
    const int N = c.N;
    if (NOBS == 0 || !(c.om > 1e-6)) return false;
    for (int e = c.lane; e < N * NOBS; e += WAVE) {
        const int k = e / L::NO, ob = e - k * L::NO;
        double dsc, dec, dsn, den;
        cbf_dist<NOBS, NMAX>(sm, c, k, ob, 0.0, dsc, dec, dsn, den);
        const int q = c.degree;
        LD(L::rdt + k * NR + 8 + NOBS + ob) = ipow_d(dsn, q) + ipow_d(den, q) - c.om * (ipow_d(dsc, q) + ipow_d(dec, q)) - c.alpha * c.cm;
    }
    SYNC();
    bool changed = false;
    for (int ob = 0; ob < c.nobs; ob++) {
        double snext = LD(L::Z + N * NZ + 6 + ob);                       // sigma_N
        for (int i = N - 1; i >= 0; i--) {
            const int j = i * NR + 8 + NOBS + ob;
            const double G = LD(L::rdt + j), push = slack_push / cbf_scale<L>(sm, i, ob);
            const double need = (snext - G + push) / c.om;
            double si = LD(L::Z + i * NZ + 6 + ob);
            if (need > si) {
                si = need;
                changed = true;
                if (c.lane == 0) {                                       // both copies of sigma_i (state of stage i, input of stage i-1)
                    LD(L::Z + i * NZ + 6 + ob) = si;
                    if (i >= 1) LD(L::Z + (i - 1) * NZ + NX + 2 + ob) = si;
                }
            }
            snext = si;
        }
    }
    SYNC();
    return changed;
}

// Crash path of the MPC-CBF NLP (crx_ipm_opts.slack_start == 2; same arithmetic as oracle/crx_oracle.c crash_point()).
// For GIVEN inputs the cheapest slacks of the rows  G_i + (1 - alpha) sigma_i - sigma_{i+1} >= 0, sigma >= 0  are the backward cascade
// sigma_N = 0, sigma_i = max(0, (sigma_{i+1} - G_i) / (1 - alpha)), and phi(u) = f(u) + w sum sigma(u) is the exact-penalty value of u.
// A crash state (ego inside, or about to enter, a safety set) is (re)started from a FEASIBLE INTERIOR point instead of u = 0, sigma = 0:
// the best of a 5 x 5 grid of constant input pairs (0.9 of the box, states inside their boxes) by phi, with its cascade pushed
// strictly inside.  One candidate per lane, 16 at a time: the lane rolls its trajectory out in registers, parks (s_k, ey_k) in the row
// arrays rt .. rtt (dead at the start; at a restart they are re-initialised right after) and walks the cascade backwards.
// crash_search() returns the winning candidate, -1 when none keeps the states inside their boxes; it runs BEFORE the interior-point
// loop, for the problems that may need it (a provable crash state, or a CBF row violated at the zero start) -- inside the (re)start loop
// its temporaries sat on top of the ~200 registers the sweeps hoist out of the loops (64 .. 164 B of scratch per lane).  crash_write()
// is the light half and is what the restart runs.
template <int NOBS, int NMAX>
__device__ __forceinline__ int crash_search(double* sm, const Ctx& c, const crx_kparams& kp) {
    using L = Lay<NOBS, NMAX>;
    // CH candidates per round, one lane each: all 25 in ONE round where their (s, ey) samples fit into the row arrays that are rewritten
    // before they are read again -- rt .. rtt (init_point), and in the full layout rsig, rw behind them (assemble_newton); two rounds of
    // 16 otherwise [r4b: the search costs a straggler of the headline batch ~2 iterations' worth, and every problem whose zero start
    // violates a row carries it]
    constexpr int NX = L::NX, NZ = L::NZ, G1 = 5;
    (void)NX; (void)NZ;
    constexpr int SPAN = (L::SLIM ? 5 : 7) * L::MR;
    constexpr int CH = 32 * 2 * (NMAX + 1) <= SPAN ? 32 : 16;
    static_assert(L::rnu == L::rt + L::MR && L::rtt == L::rt + 4 * L::MR && (L::SLIM || L::rw == L::rt + 6 * L::MR), "rt, rnu, rc, rdt, rtt, rsig, rw are contiguous");
    static_assert(NOBS == 0 || CH * 2 * (NMAX + 1) <= SPAN, "the candidates' (s, ey) samples fit in the row arrays");
    if (NOBS == 0) return -1;
    const int N = c.N, N1 = N + 1, lane = c.lane;
    double* scr = sm + L::rt;
    const int q = c.degree;
    double best = INFINITY;
    int bestc = -1;
    for (int r = 0; r * CH < G1 * G1; r++) {
        const int cand = r * CH + lane;
        const int a = cand / G1, b = cand - a * G1;
        const double u0 = 0.9 * (2.0 * a / (G1 - 1) - 1.0) * kp.delta_max, u1 = 0.9 * (2.0 * b / (G1 - 1) - 1.0) * kp.a_max;
        double val = INFINITY;
        if (lane < CH && cand < G1 * G1) {
            double x[6], v = 0.0;
            bool inside = true;
#pragma unroll
            for (int i = 0; i < 6; i++) x[i] = LD(L::Z + i);
            scr[0 * CH + lane] = x[4]; scr[1 * CH + lane] = x[5];
            // (no unrolling of the stage loops of this function: with a compile-time horizon they unroll fully, the candidates' live
            // values pile up and the register allocator spills 512 B per lane in <1,12,6,10> -- outside the iteration, but a scratch frame)
#pragma unroll 1
            for (int k = 1; k <= N; k++) {
                double xn[6];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    double t = 0.0;
#pragma unroll
                    for (int j = 0; j < 6; j++) t += kp.A[i * 6 + j] * x[j];
                    t += kp.B[i * 2] * u0 + kp.B[i * 2 + 1] * u1;
                    xn[i] = t;
                }
#pragma unroll
                for (int i = 0; i < 6; i++) x[i] = xn[i];
                inside = inside && x[0] > kp.v_min + 1e-3 && x[0] < kp.v_max - 1e-3 && x[5] > -kp.ey_max + 1e-3 && x[5] < kp.ey_max - 1e-3;
#pragma unroll
                for (int i = 0; i < 6; i++) { const double e = x[i] - LD(L::xr + k * 6 + i); v += LD(L::cst + i) * e * e; }
                scr[(2 * k) * CH + lane] = x[4]; scr[(2 * k + 1) * CH + lane] = x[5];
            }
#pragma unroll 1
            for (int k = 0; k < N; k++) v += kp.wr[0] * u0 * u0 + kp.wr[1] * u1 * u1;
            double casc = 0.0;
#pragma unroll 1
            for (int ob = 0; ob < c.nobs; ob++) {
                const double rLs = LD(L::cst + 16 + ob), rWs = LD(L::cst + 16 + L::NO + ob), lo = LD(L::cst + 8 + ob);
                double snext = 0.0;
#pragma unroll 1
                for (int i = N - 1; i >= 0; i--) {
                    const double dsc = (scr[(2 * i) * CH + lane] - LD(L::obs_s + ob * N1 + i) - lo) * rLs;
                    const double dec = (scr[(2 * i + 1) * CH + lane] - LD(L::obs_e + ob * N1 + i)) * rWs;
                    const double dsn = (scr[(2 * i + 2) * CH + lane] - LD(L::obs_s + ob * N1 + i + 1)) * rLs;
                    const double den = (scr[(2 * i + 3) * CH + lane] - LD(L::obs_e + ob * N1 + i + 1)) * rWs;
                    const double G = ipow_d(dsn, q) + ipow_d(den, q) - c.om * (ipow_d(dsc, q) + ipow_d(dec, q)) - c.alpha * c.cm;
                    double si = (snext - G) / c.om;
                    si = si < 0.0 ? 0.0 : si;
                    casc += si;
                    snext = si;
                }
            }
            v += c.wsig * casc;
            val = inside ? v : INFINITY;
        }
        const double vm = wave_min(val);
        if (vm < best) {                                  // first minimum wins (candidates in ascending order)
            const unsigned long long hit = __ballot(val == vm);
            bestc = r * CH + (__ffsll((long long)hit) - 1);
            best = vm;
        }
    }
    return bestc;
}
// ... and the point itself for candidate `bestc` of crash_search(): inputs, rolled-out states, the cascade pushed inside
template <int NOBS, int NMAX, bool cascade>
__device__ __forceinline__ void crash_write(double* sm, const Ctx& c, const crx_kparams& kp, int bestc) {
    using L = Lay<NOBS, NMAX>;
    constexpr int NX = L::NX, NZ = L::NZ, G1 = 5;
    if (NOBS == 0) return;
    const int N = c.N, lane = c.lane, q = c.degree;
    const int a = bestc / G1, b = bestc - a * G1;
    const double u0 = 0.9 * (2.0 * a / (G1 - 1) - 1.0) * kp.delta_max, u1 = 0.9 * (2.0 * b / (G1 - 1) - 1.0) * kp.a_max;
    SYNC();
    for (int e = lane; e < N * 2; e += WAVE) LD(L::Z + (e >> 1) * NZ + NX + (e & 1)) = (e & 1) ? u1 : u0;
    SYNC();
    for (int k = 0; k < N; k++) {
        if (lane < 6) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) t += LD(L::M + lane * NZ + j) * LD(L::Z + k * NZ + j);
            t += LD(L::M + lane * NZ + NX) * u0 + LD(L::M + lane * NZ + NX + 1) * u1;
            LD(L::Z + (k + 1) * NZ + lane) = t;
        }
        SYNC();
    }
    if (lane < c.nobs) {
        const int ob = lane;
        const double push = kp.opts.slack_push;
        double snext = push;
        LD(L::Z + N * NZ + 6 + ob) = push;
        LD(L::Z + (N - 1) * NZ + NX + 2 + ob) = push;
        for (int i = N - 1; i >= 0; i--) {
            double si = push;
            if constexpr (cascade) {    // (the restart leaves the cascade to restore_slacks(), which it shares with the closed-form restoration)
                double dsc, dec, dsn, den;
                cbf_dist<NOBS, NMAX>(sm, c, i, ob, 0.0, dsc, dec, dsn, den);
                const double G = ipow_d(dsn, q) + ipow_d(den, q) - c.om * (ipow_d(dsc, q) + ipow_d(dec, q)) - c.alpha * c.cm;
                si = (snext - G + push) / c.om;
                si = si < push ? push : si;
            }
            LD(L::Z + i * NZ + 6 + ob) = si;                                   // both copies of sigma_i (state of stage i, input of stage i-1)
            if (i >= 1) LD(L::Z + (i - 1) * NZ + NX + 2 + ob) = si;
            snext = si;
        }
    }
    SYNC();
}

// The planner's answer for a region whose QP has no solution: the reference's fall-back trajectory
// (overtake_traj_planner.py:365-374), zero inputs, cost +inf.
// The reference builds it from the start-line-WRAPPED ego state xcurv_ego (:366-369) although the QP's x_0 is the raw
// ego.xcurv (quirk Q5).  Only s differs between the two, and the wrapped s is the first Bezier control point
// (planner_helper.py:49; the Bernstein weights at t = 0 are exactly 1, 0, 0, 0), so it is read from there.
template <class L>
__device__ __forceinline__ void planner_fallback(double* sm, const crx_kparams& kp, int b, int lane, int N) {
    double* Xb = kp.X + (size_t)b * (N + 1) * 6;
    double* Ub = kp.U + (size_t)b * N * 2;
    const double vx0 = kp.x0[(size_t)b * 6];
    double* bs = sm + L::hg;
    double* be = sm + L::Hd;
    for (int j = lane; j <= N; j += WAVE) {
        bs[j] = kp.bez_s[(size_t)b * (N + 1) + j];
        be[j] = kp.bez_ey[(size_t)b * (N + 1) + j];
    }
    SYNC();
    const double s0 = bs[0];
    for (int e = lane; e < (N + 1) * 6; e += WAVE) {
        const int j = e / 6, i = e - j * 6;
        const double st = s0 + kp.fallback_gain * j * kp.dt_ref * vx0;
        double v = 0.0;
        if (i == 0) v = kp.fallback_gain * vx0;
        else if (i == 4) v = st;
        else if (i == 5) v = interp_lin(bs, be, N + 1, fmin(fmax(st, bs[0]), bs[N]));
        Xb[e] = v;
    }
    for (int e = lane; e < N * 2; e += WAVE) Ub[e] = 0.0;
    if (lane == 0) kp.cost[b] = INFINITY;
}

// ------------------------------------------------------------------------------------------------
// (5) the solver kernel
// ------------------------------------------------------------------------------------------------
// (No amdgpu_waves_per_eu budget.  Capping an instantiation at the register count its LDS footprint would admit --
// 168 VGPRs for <0,12>, 256 for <2,12> -- buys a resident wave per SIMD but costs spills inside the interior-point loop:
// measured -4 % on cfg3 and -2 % on cfg2 at the BASELINE batches, +1 % / +7 % only for batches of 16k planner QPs /
// 16k two-car races.  Residency is therefore min(LDS, 512 / VGPRs per SIMD); crx_debug_resident_per_cu asks the runtime.)
// Resident waves per SIMD the register allocator must leave room for.  The planner instantiation is pinned at CRX_PLANNER_WAVES
// (below); of the obstacle instantiations only the 1-obstacle, N <= 12 one (BASELINE configs[1], the MPC-CBF races) is pinned: it sat at 255 registers = 2 waves per SIMD before the restoration code was
// added and at 262 after; the bound makes the allocator park the handful of extra values (used outside the interior-
// point loop) instead of silently halving the residency.  Every other instantiation is left alone (capping those was
// measured in round 1 and rejected: spills inside the loop).
#ifndef CRX_PLANNER_WAVES
// Planner instantiation <0,12>: 3 = a third resident wave per SIMD (193 -> 168 registers, 21 dwords parked in scratch), which
// lifts the residency from 8 (register-bound) to 11 problems per CU (LDS-bound).  Measured round 2 (tools/ab_planner_waves.sh):
// 4096 QPs +0.4 %, 65536 QPs (the cfg5 shard) +10.6 %.  (Round 1 had measured -4 % at 4096 with the allocator's choice of
// spills then; make EXTRA=-DCRX_PLANNER_WAVES=1 restores the uncapped build.)
#define CRX_PLANNER_WAVES 3
#endif
#ifndef CRX_OBS1_WAVES
#define CRX_OBS1_WAVES 2
#endif
#ifndef CRX_KERNEL_EXTRA_ATTR
#define CRX_KERNEL_EXTRA_ATTR
#endif
#ifndef CRX_GEN_WAVES
#define CRX_GEN_WAVES 0   // > 0: every instantiation of this translation unit is held to that many waves per SIMD (crx_kernels_gen.hip: 2 = 256 registers, no AGPRs)
#endif
template <int NOBS, int NMAX> struct MinWaves {
    static constexpr int v = CRX_GEN_WAVES ? CRX_GEN_WAVES : (((NOBS == 1 || (CRX_W2_FLOOR && NOBS == 2)) && NMAX == 12) ? CRX_OBS1_WAVES : ((NOBS == 0 && NMAX == 12) ? CRX_PLANNER_WAVES : 1));
};

// DEG [r3]: the exponent of the super-ellipse as a compile-time constant (6 = the reference's literal, control.py:528 / :312; 0 = read
// kp.degree).  Every device function is inlined into the kernel and reads the exponent from the context, so with DEG = 6 the
// select chains of ipow_d (~18 instructions per call, 4..12 calls per CBF row evaluation) fold into the two or three products they
// stand for -- same association, same bits, no branch in the row passes (a run-time `if (p == 6)` there was measured 3 % SLOWER).
// NFIX [r3]: the horizon as a compile-time constant (10, 12 or 20: the reference's defaults and the BASELINE configs; 0 = read kp.N).
// The trip counts of the stage loops, the number of row passes (m = N NR + NOBS rows over 64 lanes) and the stage addresses become
// immediates: cfg2 0.942 -> 0.868 ms per 256 NLPs (+8.6 %), cfg3 +8 %, the cfg5 shard +6 %, cfg4 +3.7 % (tools/gpu_round3_aa.sh), same
// operations in the same order.
template <int NFIX> struct SweepUnroll { static constexpr int v = NFIX == 0 ? 1 : (CRX_SWEEP_UNROLL ? NFIX : 2); };   // forward / adjoint sweeps
template <int NOBS, int NFIX> struct RicUnroll { static constexpr int v = NFIX == 0 ? 1 : (CRX_RIC_UNROLL == 0 ? NFIX : CRX_RIC_UNROLL); };
// SPEC [r6]: 1 = TWO waves per problem (a 128-thread workgroup).  Wave 0 runs the solve as the one-wave kernel does; wave 1 sleeps at a workgroup
// barrier and, at every Newton system, factorises the reduced Hessian with the NEXT entry of the inertia-correction schedule (the convexified matrix
// on the crash path, then IPOPT's delta_w sequence) in its own work arrays while wave 0 tries the current one; the first success in schedule order
// is taken, so the iterates are those of the same kernel running the schedule one attempt at a time, bit for bit (the test compares exactly that).
// An EXPERIMENT, opt-in (crx_debug_speculation), built because VERDICT r5 asked what the idle SIMDs of a 256-problem launch could do: measured, a
// doomed attempt is cheap (the recursion stops at the first non-positive pivot: 1.7 us against 5.8 for a sweep), the longest healthy solve of the
// headline batch gains 2.5 % alone and the launch loses 1 % to the two barriers per iteration (profiles/r06_speculation.txt).
// QPM [r6]: the method of the all-linear problems (NOBS == 0; crx_ipm_opts.qp_method): 0 = Mehrotra's predictor-corrector, 1 = IPOPT's monotone
// barrier + filter line search (what every problem ran up to libcrx 0.3; instantiations with obstacles ignore it).
template <int NOBS, int NMAX, int DEG = 0, int NFIX = 0, int SPEC = 0, int QPM = 0>
__global__ void __launch_bounds__(WAVE * (1 + SPEC)) __attribute__((amdgpu_waves_per_eu(MinWaves<NOBS, NMAX>::v))) CRX_KERNEL_EXTRA_ATTR
crx_solve_kernel(const crx_kparams kp) {
    static_assert(NFIX <= NMAX, "fixed horizon inside the layout");
    using L = Lay<NOBS, NMAX>;
    static_assert(!SPEC || (NOBS > 0 && !L::SLIM && CRX_STATIC_LDS), "the speculating wave exists for the obstacle instantiations of the full, static layout");
    constexpr int NX = L::NX, NZ = L::NZ, NR = L::NR;
#if CRX_STATIC_LDS
    // [r4] The layout is a compile-time constant, so the array is STATIC: the compiler then knows its address (0) and folds it into the
    // offset fields.  As `extern __shared__` the base stays a symbol until emission and every address computed at run time carries an
    // `add ..., 0` (204 of them in <1,12,6,12>, 11 in each Riccati stage: bound by instruction issue, they cost like any other).
    __shared__ __attribute__((aligned(16))) double sm[((SPEC ? L::BYTES_SPEC : L::BYTES) + 7) / 8];
#else
    extern __shared__ __attribute__((aligned(16))) double sm[];
#endif
    int* si = (int*)(sm + L::END_D);
    const int lane = SPEC ? (int)(threadIdx.x & (WAVE - 1)) : (int)threadIdx.x, N = NFIX ? NFIX : kp.N;
    if ((int)blockIdx.x >= kp.batch) return;
    // dispatch order [r3]: workgroups start in launch order, so a caller that knows which problems are long (the iteration counts of
    // the previous control step) lists them first and the launch does not end waiting for a straggler that started last
    const int b = kp.order ? min(max(kp.order[blockIdx.x], 0), kp.batch - 1) : (int)blockIdx.x;
    if (kp.active && kp.active[kp.active_div > 1 ? b / kp.active_div : b] == 0) {   // masked launch: this problem is not part of it
        if (lane == 0) { kp.status[b] = CRX_SKIPPED; kp.iters[b] = 0; }
        return;
    }
    // [r3] Reachability screen of the planner QP, before anything is set up.  41 % of the regions of the BASELINE draw ask for a
    // lateral offset the bicycle cannot reach (~0.25 m of authority over 1 s): ey_j = e_ey' A^j x0 + sum_m e_ey' A^m B
    // u_{j-1-m}, so with |delta| <= delta_max, |a| <= a_max NO input sequence moves ey_j further than reach_gain[j] from its free
    // response reach_row[j] . x0 -- whatever the other rows do (dropping them only enlarges the feasible set).  A bound on ey_j
    // outside that interval by more than 1e-6 (the violation at which a failed solve is called infeasible) is a PROOF of
    // infeasibility: every infeasible region of the draw ends here, for one dot product per stage (the multiplier certificate of
    // DESIGN.md 4.3 needed the set-up and 2.7 iterations on them); the oracle applies the same test, so verdicts and iteration
    // counts (0 for these) stay equal.  Answer = the reference's for a failed solve: fall-back trajectory, CRX_INFEASIBLE.
    if (NOBS == 0 && kp.mode == 0 && kp.reach_screen) {
        bool out = false;
        for (int j = lane; j < N; j += WAVE) {
            double fr = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) fr += kp.reach_row[j][i] * kp.x0[(size_t)b * 6 + i];
            const double g = kp.reach_gain[j];
            out = out || kp.ey_lb[(size_t)b * N + j] > fr + g + 1e-6 || kp.ey_ub[b] < fr - g - 1e-6;
        }
        if (__ballot(out) != 0ull) {
            planner_fallback<L>(sm, kp, b, lane, N);
            if (lane == 0) { kp.status[b] = CRX_INFEASIBLE; kp.kkt[b] = INFINITY; kp.iters[b] = 0; }
            return;
        }
    }
    if constexpr (SPEC) {
        if (threadIdx.x >= WAVE) {
            // the speculating wave: parked at the workgroup barrier until wave 0 posts a command -- 0: the solve is over; 1: factorise with
            // (ctl[1] = delta_w, ctl[2] = convexified) in the second set of work arrays and post the verdict in ctl[3]; 2: nothing to try this round
            Ctx c1 = {};
            c1.N = N; c1.lane = lane;
            for (;;) {
                __syncthreads();
                const int cmd = __builtin_amdgcn_readfirstlane((int)LD(L::ctl));
                if (cmd == 0) break;
                bool ok1 = false;
                if (cmd == 1) ok1 = riccati_backward<NOBS, NMAX, RicUnroll<NOBS, NFIX>::v, 1, true>(sm, si, c1, LD(L::ctl + 1), nullptr, LD(L::ctl + 2) != 0.0);
                if (lane == 0) LD(L::ctl + 3) = ok1 ? 1.0 : 0.0;
                __syncthreads();
            }
            return;
        }
    }
    if (kp.poison) {   // diagnostics (crx_debug_poison_lds): any read of LDS this kernel did not write turns into NaN
        for (int e = lane; e < (int)(L::BYTES / 8); e += WAVE) sm[e] = __longlong_as_double(0x7ff8dead0000beefLL);
        SYNC();
    }
    Ctx c;
    c.N = N; c.lane = lane; c.m = N * NR + NOBS; c.nobs = 0;
    const int m = c.m;
    c.alpha = kp.alpha; c.om = 1.0 - kp.alpha; c.cm = 1.0 + kp.margin;
    c.degree = DEG ? DEG : kp.degree; c.wsig = kp.w_slack; c.lin_sN = 0.0; c.cconst = 0.0;
    c.b_d = kp.delta_max; c.b_a = kp.a_max; c.b_vlo = kp.v_min; c.b_vhi = kp.v_max; c.b_e = kp.ey_max;

    // ---- (3) set-up: one coalesced pass over this problem's inputs -----------------------------------
    for (int e = lane; e < NX * NZ; e += WAVE) {
        const int i = e / NZ, a = e - i * NZ;
        double v = 0.0;
        if (i < 6) {
            if (a < 6) v = kp.A[i * 6 + a];
            else if (a >= NX && a < NX + 2) v = kp.B[i * 2 + (a - NX)];
        } else if (a == NX + 2 + (i - 6)) v = 1.0;
        LD(L::M + e) = v;
    }
    for (int e = lane; e < (N + 1) * NZ; e += WAVE) {
        LD(L::Z + e) = 0.0; LD(L::dZ + e) = 0.0;
        if constexpr (!L::SLIM) { VROW(si)[L::vlo + e] = -1; VROW(si)[L::vhi + e] = -1; }
    }
    if (lane < 16) {
        double v = 0.0;
        if (lane < 6) v = kp.wq[lane];
        else if (lane < 8) v = kp.wr[lane - 6];
        LD(L::cst + lane) = v;
    }
    {   // index tables of the Riccati sweep (decoded once; the sweep itself is branch-free)
        constexpr int NTRI = (NZ * NZ <= WAVE || L::SLIM) ? 0 : NZ * (NZ + 1) / 2;   // table only when H needs the triangle form (slim: tri_decode)
        for (int e = lane; e < NTRI; e += WAVE) {
            int r = 0;
            while ((r + 1) * (r + 2) / 2 <= e) r++;
            si[L::triH + e] = (r << 8) | (e - r * (r + 1) / 2);
        }
        constexpr int NP = NX * (NX + 1) / 2 + NX;
        for (int l = lane; l < WAVE * L::UCNT; l += WAVE) {
            int i = 0, j = l - NP;
            if (ROWDPP<L> && L::UCNT == 1) {   // gradient column (i, NX) in lanes i < NX (riccati_backward keeps p there), then the pairs i <= j < NX
                if (l < NX) { i = l; j = NX; }
                else if (l < NP) {
                    int rem = l - NX;
                    while (rem >= NX - i) { rem -= NX - i; i++; }
                    j = i + rem;
                }
            } else if (l < NP) {               // pairs i <= j, j in [0, NX]
                int rem = l;
                while (rem >= NX + 1 - i) { rem -= NX + 1 - i; i++; }
                j = i + rem;
            }
            if (j < 0 || j > NX) j = 0;
            SH16(si)[L::updP + l] = (unsigned short)((i << 8) | j);
        }
    }
    SYNC();
    if (lane < 6) LD(L::Z + lane) = kp.x0[(size_t)b * 6 + lane];
    SYNC();
    // starting point: u = 0, sigma = 0, x by roll-out
    for (int k = 0; k < N; k++) {
        if (lane < 6) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) s += LD(L::M + lane * NZ + j) * LD(L::Z + k * NZ + j);
            LD(L::Z + (k + 1) * NZ + lane) = s;
        }
        SYNC();
    }
    int infeas0 = 0;
    if (kp.mode == 0) {
        // planner front-end (overtake_traj_planner.py:266-334); bez arrays staged in hg/Hd scratch
        double* bs = sm + L::hg;
        double* be = sm + L::Hd;
        for (int j = lane; j <= N; j += WAVE) {
            bs[j] = kp.bez_s[(size_t)b * (N + 1) + j];
            be[j] = kp.bez_ey[(size_t)b * (N + 1) + j];
        }
        SYNC();
        const double s0 = LD(L::Z + 4), vx0 = LD(L::Z + 0);
        for (int j = lane; j <= N; j += WAVE) {
            double st = s0 + 1.0 * j * vx0 * kp.dt_ref;                                   // :330
            st = fmin(fmax(st, bs[0]), bs[N]);                                             // :331
#pragma unroll
            for (int i = 0; i < 6; i++) LD(L::xr + j * 6 + i) = 0.0;
            LD(L::xr + j * 6 + 4) = st;
            LD(L::xr + j * 6 + 5) = interp_lin(bs, be, N + 1, st);                         // :332
            if (j < N) LD(L::wc + j) = (j >= 1 && j <= N - 2) ? kp.w_dey : 0.0;            // :325-327
        }
        c.lin_sN = -kp.w_prog;                                                             // :328
        c.cconst = kp.w_prog * s0;
        const double e0 = LD(L::Z + 5);
        if (e0 < kp.ey_lb[(size_t)b * N] - kp.opts.tol || e0 > kp.ey_ub[b] + kp.opts.tol) infeas0 = 1;
    } else {
        c.nobs = kp.n_obs ? min(max(kp.n_obs[b], 0), min(NOBS, kp.n_obs_max)) : NOBS;   // clamp: device-resident counts cannot be validated on the host
        for (int e = lane; e < (N + 1) * 6; e += WAVE)
            LD(L::xr + e) = kp.per_stage_target ? kp.xt[(size_t)b * (N + 1) * 6 + e] : kp.xt[(size_t)b * 6 + (e % 6)];
        for (int j = lane; j < N; j += WAVE) LD(L::wc + j) = 0.0;
        if (NOBS) {
            for (int e = lane; e < NOBS * (N + 1); e += WAVE) {
                const int o = e / (N + 1);
                const bool on = o < c.nobs;
                LD(L::obs_s + e) = on ? kp.obs_s[((size_t)b * kp.n_obs_max) * (N + 1) + e] : 0.0;
                LD(L::obs_e + e) = on ? kp.obs_ey[((size_t)b * kp.n_obs_max) * (N + 1) + e] : 0.0;
            }
            if (lane < NOBS) {
                LD(L::cst + 8 + lane) = lane < c.nobs ? kp.lap_off[(size_t)b * kp.n_obs_max + lane] : 0.0;
                // obstacle dimensions: per problem and obstacle slot if the caller gave them, else the descriptor's pair
                const bool own = kp.obs_dims != nullptr && lane < c.nobs;
                double ls = own ? kp.obs_dims[((size_t)b * kp.n_obs_max + lane) * 2] : kp.l_sum;
                double ws = own ? kp.obs_dims[((size_t)b * kp.n_obs_max + lane) * 2 + 1] : kp.w_sum;
                // device-resident dimensions cannot be validated on the host (the host-pointer entry point rejects them): a
                // non-positive or non-finite entry falls back to the descriptor's pair instead of turning the rows into inf / NaN
                if (!(ls > 0.0) || !isfinite(ls)) ls = kp.l_sum;
                if (!(ws > 0.0) || !isfinite(ws)) ws = kp.w_sum;
                LD(L::cst + 16 + lane) = 1.0 / ls;
                LD(L::cst + 16 + NOBS + lane) = 1.0 / ws;
            }
        }
        const double v0 = LD(L::Z + 0), e0 = LD(L::Z + 5);
        if (v0 < kp.v_min - kp.opts.tol || v0 > kp.v_max + kp.opts.tol || e0 < -kp.ey_max - kp.opts.tol ||
            e0 > kp.ey_max + kp.opts.tol)
            infeas0 = 1;                                                                   // quirk Q9
    }
    SYNC();
    // row tables: coordinate index, sign, bound, presence; and the reverse map coordinate -> rows
    for (int j = lane; j < m; j += WAVE) {
        int iv = 0;
        double sg = 0.0, bd = 0.0, on = 0.0;
        bool simple = true;
        if (j >= N * NR) {                          // sigma_0^o >= 0
            const int o = j - N * NR;
            iv = 6 + o; sg = 1.0; bd = 0.0; on = (o < c.nobs) ? 1.0 : 0.0;
        } else {
            const int k = j / NR, r = j - k * NR;
            if (r < 4) {                            // input box (control.py:572-576 / planner :280-284)
                const int i = r >> 1;
                iv = k * NZ + NX + i;
                sg = (r & 1) ? -1.0 : 1.0;
                bd = (i == 0 ? kp.delta_max : kp.a_max) * ((r & 1) ? 1.0 : -1.0);
                on = 1.0;
            } else if (r < 8) {                     // box of x_{k+1}: vx (slots 4,5), ey (slots 6,7)
                const int comp = (r < 6) ? 0 : 5;
                iv = (k + 1) * NZ + comp;
                sg = (r & 1) ? -1.0 : 1.0;
                if (kp.mode == 0) {                 // planner: vx_{k+1} <= 5 (:276); ey box for k+1 < N (:277-324)
                    if (r == 5) { bd = kp.v_max; on = 1.0; }
                    else if (r == 6 && k + 1 < N) { bd = kp.ey_lb[(size_t)b * N + k + 1]; on = 1.0; }
                    else if (r == 7 && k + 1 < N) { bd = kp.ey_ub[b]; on = 1.0; }
                } else {                            // control.py:582-586
                    bd = (r == 4) ? kp.v_min : (r == 5) ? kp.v_max : (r == 6) ? -kp.ey_max : kp.ey_max;
                    on = 1.0;
                }
                if (!isfinite(bd)) on = 0.0;
            } else if (r < 8 + NOBS) {              // sigma_{k+1}^o >= 0 (control.py:559-561)
                const int o = r - 8;
                iv = k * NZ + NX + 2 + o; sg = 1.0; bd = 0.0; on = (o < c.nobs) ? 1.0 : 0.0;
            } else {                                // CBF row: evaluated, not table-driven
                simple = false;
                on = ((r - 8 - NOBS) < c.nobs) ? 1.0 : 0.0;
            }
        }
        if constexpr (!L::SLIM) {
            if (simple && on != 0.0) {
                if (sg > 0.0) VROW(si)[L::vlo + iv] = (typename L::vrow_t)j; else VROW(si)[L::vhi + iv] = (typename L::vrow_t)j;
            }
        }
        if (on == 0.0 || !simple) { sg = 0.0; bd = 0.0; iv = 0; }
        SH16(si)[L::riv + j] = (unsigned short)(iv | (sg != 0.0 ? RIV_SIMPLE : 0) | (sg < 0.0 ? RIV_NEG : 0));
        if constexpr (L::SLIM) {
            if (!simple) { const int k = j / NR; LD(L::csc + k * L::NO + (j - k * NR - 8 - NOBS)) = on; }
        } else {
            LD(L::rb + j) = bd;
            LD(L::rsc + j) = on;
            LD(L::rsig + j) = 0.0; LD(L::rw + j) = 0.0;
        }
        LD(L::rnu + j) = on;       // multiplier start 1 (0 for absent rows)
        LD(L::rt + j) = 1.0;
        LD(L::rc + j) = 1.0;
        LD(L::rdt + j) = 0.0; LD(L::rtt + j) = 1.0;
    }
    SYNC();
    const crx_ipm_opts o = kp.opts;
    // [r3] Provable lower bounds of the slacks.  IPOPT starts every sigma at its bound (pushed to 1e-2); a car that starts inside an
    // obstacle's safety set needs slacks of 1e2..1e5 (each row i needs sigma_i >= (sigma_{i+1} - G_i) / (1 - alpha): a 1/(1-alpha) growth
    // over the stages the car cannot leave the set in).  What CAN be said before solving: s_k and ey_k stay within reach_s[k],
    // reach_gain[k] of the free response (boxed inputs), so G_i = g(x_{i+1}) - (1 - alpha) g(x_i) - alpha (1 + margin) has an upper
    // bound Gmax_i over ALL admissible inputs, and backwards from L_N = 0, L_i = max(0, (L_{i+1} - Gmax_i) / (1 - alpha)) is a
    // PROVABLE lower bound of sigma_i at any feasible point.  slack_start == 1 (libcrx 0.1.3's option) starts the slacks there;
    // slack_start == 2 (default, [r4]) only asks WHETHER some L_i is positive -- a crash state -- and then takes the crash path
    // (crash_point() above).  Zero, i.e. nothing changes, for every problem whose rows can be met without slack.  Same arithmetic as
    // oracle/crx_oracle.c slack_lower_bounds().
    int crash_state = 0;
    if (NOBS && kp.slack_start && c.om > 1e-6) {
        bool any = false;
        if (lane < c.nobs) {
            const int ob = lane, q = c.degree;
            double Lb = 0.0;
            for (int i = N - 1; i >= 0; i--) {
                double dsc, dec, dsn, den;
                cbf_dist<NOBS, NMAX>(sm, c, i, ob, 0.0, dsc, dec, dsn, den);
                const double rLs = LD(L::cst + 16 + ob), rWs = LD(L::cst + 16 + L::NO + ob);
                const double rsc = kp.reach_s[i] * rLs, rec = kp.reach_gain[i] * rWs, rsn = kp.reach_s[i + 1] * rLs, ren = kp.reach_gain[i + 1] * rWs;
                const double mx_sn = fmax(fabs(dsn - rsn), fabs(dsn + rsn)), mx_en = fmax(fabs(den - ren), fabs(den + ren));
                const double mn_sc = fabs(dsc) > rsc ? fabs(dsc) - rsc : 0.0, mn_ec = fabs(dec) > rec ? fabs(dec) - rec : 0.0;
                const double Gmax = ipow_d(mx_sn, q) + ipow_d(mx_en, q) - c.om * (ipow_d(mn_sc, q) + ipow_d(mn_ec, q)) - c.alpha * c.cm;
                Lb = (Lb - Gmax) / c.om;
                Lb = Lb > 0.0 ? Lb : 0.0;
                if (Lb > 0.0) {
                    any = true;
                    if (kp.slack_start == 1) {                          // both copies of sigma_i (state of stage i, input of stage i-1)
                        LD(L::Z + i * NZ + 6 + ob) = Lb;
                        if (i >= 1) LD(L::Z + (i - 1) * NZ + NX + 2 + ob) = Lb;
                    }
                }
            }
        }
        crash_state = __ballot(any) != 0ull;
        SYNC();
    }
    // [r4] crash path (include/crx.h crx_ipm_opts.slack_start == 2): a provable crash state starts from the feasible interior point of
    // crash_point(); a solve that started at zero and stalls on violated CBF rows restarts ONCE from such a point.  `crash` = this
    // solve is on the crash path (it also selects the convexified inertia retry below).
    const bool crash_path = NOBS > 0 && kp.slack_start >= 2 && o.restore_iters >= 0 && c.om > 1e-6 && c.nobs > 0;
    int crash = 0;
    double mu = o.mu_init, dw_last = 0.0, E0 = INFINITY, theta_min = 0.0, theta_max = INFINITY;
    double f = 0.0;
    int nf = 0, status = 1, it = 0;
    double mact = 0.0;
    for (int j = lane; j < m; j += WAVE) mact += (LD(L::rnu + j) != 0.0) ? 1.0 : 0.0;   // rnu = presence flag of the row (table loop above)
    mact = wave_sum(mact);
    const double kappa_sigma = 1e10, smax = 100.0, eta = 1e-8;
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    // row statistics of the current iterate: sum nu, max |c - t|, max and min of t*nu over the present rows.  Computed
    // here for the start point and afterwards inside the accept pass, which touches every row anyway.
    // logsum_t = sum_j log t_j over the present rows, the barrier term of the merit function at the iterate: taken over
    // from the accepted line-search trial (which computed it for exactly the slacks that become the iterate) instead of
    // being recomputed in the row-step pass of the next iteration -- two wave reductions, a log and a frexp per row less.
    double nus = 0.0, e_p = 0.0, cmax = 0.0, cmin = INFINITY, logsum_t = 0.0;
    auto row_stats = [&]() {
        nus = 0.0; e_p = 0.0; cmax = 0.0; cmin = INFINITY;
        LogAcc lgs;
        for (int j = lane; j < m; j += WAVE) {
            const bool on = row_scale<L>(sm, si, j, N) != 0.0;
            const double t = LD(L::rt + j), nu = LD(L::rnu + j);
            nus += nu;
            e_p = fmax(e_p, on ? fabs(LD(L::rc + j) - t) : 0.0);
            cmax = fmax(cmax, on ? t * nu : 0.0);
            cmin = fmin(cmin, on ? t * nu : INFINITY);
            lgs.mul(on ? t : 1.0);
        }
        nus = wave_sum(nus); e_p = wave_max(e_p); cmax = wave_max(cmax); cmin = wave_min(cmin);
        logsum_t = lgs.wave_total();
    };
    // The interior-point loop sits inside a (re)start loop.  stage 1: take the crash point (a provable crash state at the start; a
    // stalled solve once); stage 0: slacks and multipliers at the point Z holds; then the iteration.  When the line search finds no
    // acceptable step the (out-of-loop, rare) crash restart or closed-form restoration below re-initialises and the loop is entered
    // again.  Keeping all of that outside the loop body keeps its registers out of the loop's allocation (inside, the restoration
    // cost the 1-obstacle instantiation 14 registers = one resident wave per SIMD).
    // ls_failed: 1 = no acceptable step, 2 = jam (JAM_COUNT accepted steps in a row shorter than JAM_ALPHA while the
    // constraints are still violated: the slacks of violated CBF rows are collapsing and every step is cut to nothing --
    // IPOPT's alpha < alpha_min test sends it to restoration from the same situation)
    constexpr int JAM_COUNT = 5;
    const int STALL_ITERS = o.stall_iters;   // [r6] crx_ipm_opts.stall_iters, by problem class (crx_cbf_desc_default): 50 for N <= 12 with one obstacle slot, 100 otherwise
    const double JAM_ALPHA = 1e-3;
    int n_restore = 0, ls_failed = 0, jam = 0, jam_on = (NOBS > 0 && o.restore_iters >= 0), it_limit = 0, cvx_run = 0;
    constexpr int CVX_PROBE = 4;
    int scaled = 0;
    // slacks and multipliers at the point Z holds -- the start, and again at the crash restart (every row array is rewritten:
    // crash_point parks its samples there); then the merit pieces of that point.  Two call sites, both outside the interior-point loop.
    auto init_point = [&]() {
        // CBF row scaling at the FIRST starting point (IPOPT's gradient-based scaling, measured in the reference's variables)
        if (NOBS && !scaled) {
            for (int e = lane; e < N * NOBS; e += WAVE) {
                const int k = e / NOBS, ob = e - k * NOBS;
                if (ob < c.nobs) {
                    double dsc, dec, dsn, den;
                    cbf_dist<NOBS, NMAX>(sm, c, k, ob, 0.0, dsc, dec, dsn, den);
                    const int q = c.degree;
                    double gm = 1.0;
                    const double rLs = LD(L::cst + 16 + ob), rWs = LD(L::cst + 16 + NOBS + ob);
                    gm = fmax(gm, fabs(q * ipow_d(dsn, q - 1) * rLs));
                    gm = fmax(gm, fabs(q * ipow_d(den, q - 1) * rWs));
                    if (k > 0) {
                        gm = fmax(gm, fabs(c.om * q * ipow_d(dsc, q - 1) * rLs));
                        gm = fmax(gm, fabs(c.om * q * ipow_d(dec, q - 1) * rWs));
                    }
                    LD(L::SLIM ? L::csc + k * L::NO + ob : L::rsc + k * NR + 8 + NOBS + ob) = fmin(1.0, o.grad_scale_max / gm);
                }
            }
            SYNC();
        }
        scaled = 1;
        // slacks t = max(|c|, push); multipliers 1, simple-bound rows: the reduced cost gradient that pushes against the bound
        eval_rows<NOBS, NMAX>(sm, si, c);
        SYNC();
        for (int j = lane; j < m; j += WAVE) {
            const bool on = row_scale<L>(sm, si, j, N) != 0.0;
            LD(L::rt + j) = on ? fmax(fabs(LD(L::rc + j)), o.slack_push) : 1.0;
            LD(L::rtt + j) = on ? 1.0 : 0.0;        // the multiplier start, parked while first_order runs with nu = 0
            LD(L::rnu + j) = 0.0;
            LD(L::rdt + j) = 0.0;
        }
        SYNC();
        first_order<NOBS, NMAX>(sm, si, c);          // with nu = 0: ga = grad f
        (void)dual_infeasibility<NOBS, NMAX>(sm, c);  // ga <- reduced cost gradient (inputs, sigma_0)
        for (int j = lane; j < m; j += WAVE) {
            double nu = LD(L::rtt + j);
            const int pk = RIVT(si, j);
            if (nu != 0.0 && (pk & RIV_SIMPLE)) {
                const int iv = RIV_IDX(pk);
                const int kk = iv / NZ, a = iv - kk * NZ;
                if (a >= NX || (kk == 0 && a >= 6)) {          // input or sigma_0 coordinate
                    const double gg = RIV_SGN(pk) * LD(L::ga + iv);
                    if (gg > 1.0) nu = gg;
                }
            }
            LD(L::rnu + j) = nu;
            LD(L::rtt + j) = 1.0;
        }
        SYNC();
        first_order<NOBS, NMAX>(sm, si, c);
        f = cost_value<NOBS, NMAX>(sm, c, 0.0);
        row_stats();
        mu = o.mu_init; nf = 0; dw_last = 0.0; status = 1; jam = 0;
        theta_min = -1.0; theta_max = INFINITY;   // < 0: the filter's theta_min / theta_max are taken at the next step (start, and after a restart / restoration)
    };
    // the candidate search runs here, before the loops, for every problem that may take the crash path: a provable crash state (it starts
    // from the point), or a CBF row violated at the zero start (it may stall and restart from the point)
    int crash_cand = -1;
    if (NOBS && crash_path) {
        bool viol = false;
        for (int e = lane; e < N * NOBS; e += WAVE) {
            const int k = e / L::NO, ob = e - k * L::NO;
            if (ob < c.nobs) {
                double dsc, dec, dsn, den;
                cbf_dist<NOBS, NMAX>(sm, c, k, ob, 0.0, dsc, dec, dsn, den);
                const int q = c.degree;
                viol = viol || (ipow_d(dsn, q) + ipow_d(den, q) - c.om * (ipow_d(dsc, q) + ipow_d(dec, q)) - c.alpha * c.cm < 0.0);
            }
        }
        if (crash_state || __ballot(viol) != 0ull) crash_cand = __builtin_amdgcn_readfirstlane(crash_search<NOBS, NMAX>(sm, c, kp));
        // (a solve that STARTS on the crash path has a budget too -- three times the restoration budget + 1 -- after which it ends CRX_RESTORED like
        // a restarted one: feasible through its slacks, not optimal, instead of crawling to max_iter)
        // (slack_start == 3, eager: a violated zero start is enough -- include/crx.h; the restart then never fires)
        if ((crash_state || kp.slack_start == 3) && crash_cand >= 0) { crash_write<NOBS, NMAX, true>(sm, c, kp, crash_cand); crash = 1; n_restore = 1; it_limit = 1 + 3 * o.restore_iters; }
    }
    init_point();
    if constexpr (NOBS == 0 && QPM == 0) {
    // ---- [r6] Mehrotra's predictor-corrector for the all-linear problems (planner region QPs, MPC-CBF NLPs without an obstacle slot): a convex QP has
    // ONE solution, so the path to it is free (oracle/crx_oracle.c qp_pc_solve has the algorithm note and runs the same arithmetic
    // on a dense Cholesky).  Per iteration ONE factorisation -- the Riccati sweep, with the factors kept (KEEPF) -- and two solves: the affine-scaling
    // predictor (mu = 0), sigma = (mu_aff / mu)^3, and the corrector for t nu = sigma mu - dt_aff dnu_aff through riccati_backward_vec; separate primal
    // and dual step lengths, no merit function, no filter, no line search.  Start, error measure, termination test, infeasibility proofs: as below.
    (void)theta_min; (void)theta_max; (void)nf; (void)dw_last; (void)logsum_t; (void)cmin; (void)mu; (void)crash; (void)crash_path;
    double gap = 0.0;
    for (int j = lane; j < m; j += WAVE) gap += (row_scale<L>(sm, si, j, N) != 0.0) ? LD(L::rt + j) * LD(L::rnu + j) : 0.0;
    gap = wave_sum(gap);
    constexpr int RP = (L::MR + WAVE - 1) / WAVE;
    for (;; it++) {
#if CRX_OPAQUE_LANE
        asm volatile("" : "+v"(c.lane));
        const int lane = c.lane;   // shadows the kernel's `lane` inside the loop body (see the loop below)
#endif
        const double sd = fmax(smax, nus / fmax(mact, 1.0)) / smax;
        const double e_du = dual_infeasibility<NOBS, NMAX, (SweepUnroll<NFIX>::v > 2 ? SweepUnroll<NFIX>::v : 1)>(sm, c);
        E0 = fmax(e_du / sd, fmax(e_p, cmax / sd));
        if (E0 <= o.tol && e_du <= o.dual_inf_tol && e_p <= o.constr_viol_tol && cmax <= o.compl_inf_tol) { status = 0; break; }   // IPOPT's complete test
        if (it >= o.max_iter) break;
        const double mu_k = gap / fmax(mact, 1.0);
        // ---- predictor: the affine-scaling direction ---------------------------------------------------
        assemble_newton<NOBS, NMAX>(sm, si, c, 0.0);
        if (!riccati_backward<NOBS, NMAX, RicUnroll<NOBS, NFIX>::v, 0, false, true>(sm, si, c, 0.0)) break;   // (a convex QP: cannot happen short of overflow)
        riccati_forward<NOBS, NMAX, SweepUnroll<NFIX>::v>(sm, c);
        double dta[RP], dna[RP], rpm = 0.0, rdm = 0.0;
#pragma unroll
        for (int q_ = 0; q_ < RP; q_++) {
            const bool jv = lane + q_ * WAVE < m;
            const int j = jv ? lane + q_ * WAVE : 0;
            const int pk = RIVT(si, j);
            const bool on = jv && LD(L::rsc + j) != 0.0;
            const double jd = RIV_SGN(pk) * LD(L::dZ + RIV_IDX(pk));
            const double t = LD(L::rt + j), nu = LD(L::rnu + j), rti = LD(L::rtt + j), rcj = LD(L::rc + j), rsj = LD(L::rsig + j), rwj = LD(L::rw + j);
            const double rp = rcj - t;
            dta[q_] = sel(on, jd + rp, 0.0);
            dna[q_] = sel(on, -rwj + rsj * (rp - dta[q_]), 0.0);
            rpm = fmax(rpm, -dta[q_] * rti);                  // (absent rows: dt = 0)
            rdm = fmax(rdm, sel(on, -dna[q_] * frcp(nu), 0.0));
        }
        wave_max2(rpm, rdm);
        const double apa = rpm > 1.0 ? 1.0 / rpm : 1.0, ada = rdm > 1.0 ? 1.0 / rdm : 1.0;      // to the boundary
        double ga_ = 0.0;
#pragma unroll
        for (int q_ = 0; q_ < RP; q_++) {
            const bool jv = lane + q_ * WAVE < m;
            const int j = jv ? lane + q_ * WAVE : 0;
            ga_ += sel(jv && LD(L::rsc + j) != 0.0, (LD(L::rt + j) + apa * dta[q_]) * (LD(L::rnu + j) + ada * dna[q_]), 0.0);
        }
        const double mu_aff = wave_sum(ga_) / fmax(mact, 1.0);
        double sigma = mu_aff / mu_k;
        sigma = sigma * sigma * sigma;
        // (the centering target never goes below IPOPT's smallest barrier parameter: with sigma -> 1e-9 the products t nu fall to 1e-30 in two steps and
        // the reduced gradient sits on a rounding floor above tol for ever -- oracle note)
        const double smu = fmax(sigma * mu_k, o.tol / 10.0);
        // ---- corrector: w <- w_aff - (sigma mu - dt_aff dnu_aff) / t, hg <- hg + J'(w - w_aff); same factor -----------
        // (second pass, rare: a corrected step shorter than PC_SHORT_STEP is redone as a plain centring step t nu = mu -- the second-order term was
        // computed for a full affine step that cannot be taken; oracle note.  The row / gradient updates are written as differences to what is applied.)
        constexpr double PC_SHORT_STEP = 0.2;
        double cr[RP], s_new = smu, s_old = 0.0, c_old = 0.0, a_p = 1.0, a_d = 1.0;
#pragma unroll
        for (int q_ = 0; q_ < RP; q_++) cr[q_] = dta[q_] * dna[q_];
        const double tau = fmax(o.tau_min, 1.0 - mu_k);
        for (int pass = 0;; pass++) {
            const double c_new = pass == 0 ? 1.0 : 0.0;       // weight of the second-order term
            SYNC();                                           // every lane has read its dZ entries: dZ is scratch from here to the forward sweep
#pragma unroll
            for (int q_ = 0; q_ < RP; q_++) {
                const bool jv = lane + q_ * WAVE < m;
                const int j = jv ? lane + q_ * WAVE : 0;
                const bool on = jv && LD(L::rsc + j) != 0.0;
                const double dwj = sel(on, -((s_new - c_new * cr[q_]) - (s_old - c_old * cr[q_])) * LD(L::rtt + j), 0.0);
                const double rwj = LD(L::rw + j);
                LD(SINK(jv, L::rw + j)) = rwj + dwj;
                LD(SINK(jv, L::dZ + j)) = dwj;                // (MR <= NV without obstacles: the rows fit)
            }
            SYNC();
            COORDS(e, ev, lane, N * NZ + NX) {
                const int k = e / NZ, a = e - k * NZ;
                int rl, rh;
                coord_rows<L>(si, c, e, k, a, rl, rh);
                const double dl = LD(L::dZ + (rl >= 0 ? rl : 0)), dh = LD(L::dZ + (rh >= 0 ? rh : 0)), g0 = LD(L::hg + e);
                LD(SINK(ev && !(k == N && a >= NX), L::hg + e)) = g0 + sel(rl >= 0, dl, 0.0) - sel(rh >= 0, dh, 0.0);
            }
            SYNC();
            riccati_backward_vec<NOBS, NMAX, SweepUnroll<NFIX>::v>(sm, c);
            riccati_forward<NOBS, NMAX, SweepUnroll<NFIX>::v>(sm, c);
            rpm = 0.0; rdm = 0.0;
#pragma unroll
            for (int q_ = 0; q_ < RP; q_++) {
                const bool jv = lane + q_ * WAVE < m;
                const int j = jv ? lane + q_ * WAVE : 0;
                const int pk = RIVT(si, j);
                const bool on = jv && LD(L::rsc + j) != 0.0;
                const double jd = RIV_SGN(pk) * LD(L::dZ + RIV_IDX(pk));
                const double t = LD(L::rt + j), nu = LD(L::rnu + j), rti = LD(L::rtt + j), rcj = LD(L::rc + j), rsj = LD(L::rsig + j), rwj = LD(L::rw + j);
                const double rp = rcj - t;
                dta[q_] = sel(on, jd + rp, 0.0);              // the corrected step of the row's slack / multiplier
                dna[q_] = sel(on, -rwj + rsj * (rp - dta[q_]), 0.0);
                rpm = fmax(rpm, -dta[q_] * rti);
                rdm = fmax(rdm, sel(on, -dna[q_] * frcp(nu), 0.0));
            }
            wave_max2(rpm, rdm);
            a_p = rpm > tau ? tau / rpm : 1.0; a_d = rdm > tau ? tau / rdm : 1.0;
            if (pass == 0 && fmin(a_p, a_d) < PC_SHORT_STEP) { s_old = s_new; c_old = 1.0; s_new = fmax(mu_k, o.tol / 10.0); continue; }
            break;
        }
        static_assert(L::MR <= L::NV, "row scratch inside dZ");
        if (kp.trace && b == kp.trace_problem && it < (kp.trace_rows < 0 ? -kp.trace_rows : kp.trace_rows) && lane == 0) {
            double* tr = kp.trace + (size_t)it * 16;
            tr[0] = e_du / sd; tr[1] = e_p; tr[2] = cmax / sd; tr[3] = mu_k; tr[4] = a_p; tr[5] = a_d; tr[6] = sigma; tr[7] = 1.0;
        }
        // ---- accept: no merit function, no filter (linear rows: the primal residual shrinks by 1 - a_p) ---------------
        SYNC();
        COORDS(e, ev, lane, N * NZ + NX) {
            const double zn = LD(L::Z + e) + a_p * LD(L::dZ + e);
            LD(SINK(ev, L::Z + e)) = zn;
        }
        SYNC();
        double numax = 0.0, th = 0.0, gs = 0.0;
        nus = 0.0; cmax = 0.0;
#pragma unroll
        for (int q_ = 0; q_ < RP; q_++) {
            const bool jv = lane + q_ * WAVE < m;
            const int j = jv ? lane + q_ * WAVE : 0;
            const int pk = RIVT(si, j);
            const bool on = jv && LD(L::rsc + j) != 0.0;
            const double v = RIV_SGN(pk) * (LD(L::Z + RIV_IDX(pk)) - LD(L::rb + j));
            double tn = LD(L::rt + j) + a_p * dta[q_];
            tn = sel(v > tn, v, tn);                          // the slack never lags behind its row (a full primal step makes them equal)
            const double nn = LD(L::rnu + j) + a_d * dna[q_];
            LD(SINK(on, L::rt + j)) = tn;
            LD(SINK(on, L::rnu + j)) = nn;
            LD(SINK(jv, L::rc + j)) = sel(on, v, 1.0);
            numax = fmax(numax, sel(on, nn, 0.0));
            th = fmax(th, sel(on, fabs(v - tn), 0.0));
            nus += sel(on, nn, 0.0);
            gs += sel(on, tn * nn, 0.0);
            cmax = fmax(cmax, sel(on, tn * nn, 0.0));
        }
        SYNC();
        {
            double z0 = 0.0;
            wave_max4(numax, th, cmax, z0);
            double z1 = 0.0, z2 = 0.0;
            wave_sum4(nus, gs, z1, z2);
        }
        gap = gs;
        e_p = th;
        first_order<NOBS, NMAX>(sm, si, c);
        if (numax > 1e12 && th > 1e-6) { status = CRX_STALLED; it++; break; }   // IPOPT's divergence heuristic: not a proof
        if (th > 1e-6) {   // still violated after the step: look for the proof that it must be
            if (box_certificate<NOBS, NMAX>(sm, si, c, kp.delta_max, kp.a_max) < -1e-8 * numax) { status = 2; it++; break; }
        }
    }
    f = cost_value<NOBS, NMAX>(sm, c, 0.0);
    } else {
    if (NOBS && crash) {
        // [r4] the crash start's barrier parameter comes from its own complementarity (oracle/crx_oracle.c has the note): slacks of 1e2 .. 1e5 with
        // multipliers of 1 are nowhere near the central path of mu = 0.1, and the iteration crawled for 25 steps before mu moved at all.
        // mu_0 = 0.1 x mean_j(t_j nu_j) over the present rows, not below mu_init.  (Absent rows hold nu = 0.)
        double sc = 0.0;
        for (int j = lane; j < m; j += WAVE) sc += LD(L::rt + j) * LD(L::rnu + j);
        sc = wave_sum(sc);
        mu = fmin(fmax(0.1 * sc / fmax(mact, 1.0), o.mu_init), 1e6);
    }
    for (;;) {
    ls_failed = 0;
    // Nothing is to be carried in registers from one pass to the next: without this barrier the loop-invariant operands
    // of the sweeps (model-matrix rows, lane maps: ~200 registers) are hoisted out of BOTH loops and stay live across
    // the restart / restoration code, whose own temporaries then push the kernel past 256 registers.
    asm volatile("" ::: "memory");
#if CRX_OPAQUE_OUTER
    // [r4] ... and the lane index is made opaque once per PASS of this outer loop (obstacle instantiations): what the sweeps derive
    // from it is still hoisted out of the interior-point loop (they are issue-bound: the hoisted maps are worth 4..5 %), but only to
    // the top of the pass, not out of the outer loop -- so it is dead while the crash restart at the bottom of the pass runs.
    // Without: +21 registers and 64 B of scratch per lane for <1,12>, 164 B for <3,20>.
    if (NOBS > 0) asm volatile("" : "+v"(c.lane));
    const int lane = c.lane;   // shadows the kernel's `lane` inside the pass
#endif

    for (;; it++) {
        long long tc0 = CLK();
#if CRX_OPAQUE_LANE
        // [r3] Planner instantiations: the lane index is made opaque once per iteration, so that everything the phases derive
        // from it (which entry a lane owns, LDS addresses, the selects of the row / coordinate passes) is recomputed inside
        // the iteration instead of being hoisted out of the interior-point loop and kept alive across it.  <0,12>: 168 VGPRs +
        // 116 B of scratch per lane -> 160 and NO scratch (the parked dwords were x6..10 the algorithmic HBM traffic of the
        // planner launches); cfg3 +4.7 %, cfg5 +1.7 % (tools/gpu_round3_a.sh).  The obstacle instantiations run one wave per
        // SIMD and are bound by instruction ISSUE (one instruction per four clocks and wave: profiles/r03_pmc_issue.txt); the
        // recomputation is ~19 % more VALU instructions there and costs 4..5 % (cfg2, cfg4, races) although it frees 40..140
        // registers -- they keep the hoisted maps.  (With the barrier, the full-layout 3-obstacle instantiations also produced
        // a kernel that faults on MI355X / ROCm 7.0.2 -- not understood; found by the GPU suite, tools/gpu_round3_b.sh.)
        if (NOBS == 0 || CRX_OPAQUE_LANE >= 2) asm volatile("" : "+v"(c.lane));   // (2: every instantiation -- the A/B build of the fault hunt, tools/variants.sh)
        const int lane = c.lane;   // shadows the kernel's `lane` inside the loop body
#endif
        // ---- KKT error -----------------------------------------------------------------------------
        double e_c = cmax;
        const double sd = fmax(smax, nus / fmax(mact, 1.0)) / smax;
        long long tc1 = CLK();
        const double e_du = dual_infeasibility<NOBS, NMAX, (SweepUnroll<NFIX>::v > 2 ? SweepUnroll<NFIX>::v : 1)>(sm, c);
        const double e_d = e_du / sd;
        long long tc2 = CLK();
        e_c /= sd;
        E0 = fmax(e_d, fmax(e_p, e_c));
        if (E0 <= o.tol) {
            // [r6] IPOPT's COMPLETE termination test (OptimalityErrorConvergenceCheck; the reference runs IPOPT on its defaults): the scaled
            // error <= tol AND the unscaled dual infeasibility <= dual_inf_tol (1), constraint violation <= constr_viol_tol (1e-4),
            // complementarity <= compl_inf_tol (1e-4).  Unscaled = no s_d, CBF rows in the reference's units (the simple rows are unscaled; t nu is
            // invariant under the row scaling).  The second half binds on crash states: multipliers of 1e7..1e9 make s_d 1e4..1e7 and the scaled
            // complementarity passes at mu = 1e-4 already.  Same test in oracle/crx_oracle.c.  The row pass runs on this (once-per-solve) path only.
            double vu = e_p;
            if (NOBS) {
                for (int e = lane; e < N * NOBS; e += WAVE) {
                    const int k = e / L::NO, ob = e - k * L::NO, j = k * NR + 8 + NOBS + ob;
                    const double sc = cbf_scale<L>(sm, k, ob);
                    if (sc != 0.0) vu = fmax(vu, fabs(LD(L::rc + j) - LD(L::rt + j)) / sc);
                }
                vu = wave_max(vu);
            }
            if (e_du <= o.dual_inf_tol && vu <= o.constr_viol_tol && cmax <= o.compl_inf_tol) { status = 0; break; }
        }
        if (it >= o.max_iter) break;
        if (NOBS && n_restore > 0 && it >= it_limit) { status = 3; break; }   // restoration budget used up (CRX_RESTORED)
        // ---- barrier update ------------------------------------------------------------------------
        for (;;) {
            // max_j |t_j nu_j - mu| from the extremes of t*nu: no pass over the rows
            const double e_cm = fmax(cmax - mu, mu - cmin) / sd;
            const double Emu = fmax(e_d, fmax(e_p, e_cm));
            if (Emu <= o.kappa_eps * mu && mu > o.tol / 10.0) {
                mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, o.theta_mu == 1.5 ? mu * sqrt(mu) : pow(mu, o.theta_mu)));
                nf = 0;
            } else
                break;
        }
        const double tau = fmax(o.tau_min, 1.0 - mu);
        // ---- Newton step ---------------------------------------------------------------------------
        long long tc3 = CLK();
        assemble_newton<NOBS, NMAX>(sm, si, c, mu);
        long long tc4 = CLK();
        // inertia correction: retry the sweep with a growing regularisation dw until every pivot is positive (one call
        // site: the sweep is inlined once)
        double dw = 0.0;
        long long tsub[4] = {0, 0, 0, 0};
        bool ok;
        // ... and the convexification is sticky [r4]: after an iteration that needed it the next ones START convexified, every CVX_PROBE-th of
        // such a run tries the exact matrix first again.  (The exact reduced Hessian of a crash state has the wrong inertia for a dozen
        // iterations in a row, and the sweep notices at its far end: the longest solve of the headline batch spent 17 % of its clocks on
        // attempts that were bound to fail.  Same rule in oracle/crx_oracle.c; problems off the crash path never get here.)
        int used_convex = 0;
        if (NOBS && crash && cvx_run > 0 && (cvx_run % CVX_PROBE) != 0) {
            if constexpr (!SPEC) {
                for (int k = lane; k < N; k += WAVE) { LD(L::kS + 2 * k) = 0.0; LD(L::kE + 2 * k) = 0.0; }
                SYNC();
            }
            used_convex = 1;
        }
        if constexpr (SPEC) {
            // [r6] two attempts of the schedule at a time: this wave tries S = (dw, convexified), the speculating wave S' = next(S) -- next() being
            // the sequential code below: the convexified matrix first while it is still ahead (crash path), then IPOPT's delta_w sequence.  The
            // first success in schedule order is taken; a success of S' is copied over (feedback gains, feed-forward, sigma_0 step: all the
            // forward sweep reads).  (The pre-convexified start above did not touch LDS here: `cz` carries it.)
            double sdw = 0.0;
            int cz = used_convex, avail = (crash && !used_convex) ? 1 : 0, tries = 0;
            for (;;) {
                double ndw = sdw;
                int ncz = cz, ntries = tries;
                if (avail) ncz = 1;
                else { ndw = tries == 0 ? (dw_last == 0.0 ? 1e-4 : fmax(1e-20, dw_last / 3.0)) : sdw * (dw_last == 0.0 ? 100.0 : 8.0); ntries = tries + 1; }
                const bool nvalid = !(ndw > 1e40);
                const bool post = nvalid && kp.spec_idle == 0;      // (spec_idle: diagnostics -- the second wave is never asked: the schedule runs one attempt at a time)
                if (lane == 0) { LD(L::ctl) = post ? 1.0 : 2.0; LD(L::ctl + 1) = ndw; LD(L::ctl + 2) = ncz ? 1.0 : 0.0; }
                __syncthreads();
                ok = riccati_backward<NOBS, NMAX, RicUnroll<NOBS, NFIX>::v, 0, true>(sm, si, c, sdw, tsub, cz != 0);
                __syncthreads();
                if (ok) { dw = sdw; used_convex = cz; break; }
                dw = ndw;
                if (!nvalid) break;                               // the schedule is exhausted (sequential: dw > 1e40)
                if (!post) { sdw = ndw; cz = ncz; avail = 0; tries = ntries; continue; }
                if (__builtin_amdgcn_readfirstlane((int)LD(L::ctl + 3)) != 0) {
                    for (int e = lane; e < N * L::NU * NX; e += WAVE) LD(L::Kk + e) = LD(L::Kk2 + e);
                    for (int e = lane; e < N * L::NU; e += WAVE) LD(L::kf + e) = LD(L::kf2 + e);
                    if (lane < NZ) LD(L::dZ + lane) = LD(L::dZ2 + lane);
                    SYNC();
                    ok = true; used_convex = ncz;
                    break;
                }
                // both have the wrong inertia: two steps down the schedule (after S' the convexified retry is behind us)
                sdw = ntries == 0 ? (dw_last == 0.0 ? 1e-4 : fmax(1e-20, dw_last / 3.0)) : ndw * (dw_last == 0.0 ? 100.0 : 8.0);
                cz = ncz; avail = 0; tries = ntries + 1;
                dw = sdw;
                if (sdw > 1e40) break;
            }
        } else
        for (int tries = 0, convex = (NOBS > 0 && crash && !used_convex) ? 0 : 1;; ) {
            ok = riccati_backward<NOBS, NMAX, RicUnroll<NOBS, NFIX>::v>(sm, si, c, dw, tsub);
            if (ok) break;
            if (NOBS && !convex) {
                used_convex = 1;
                // [r4] crash path (iii): first retry WITHOUT the reverse-convex part of the CBF curvature (-nu hess g_{k+1}, the kS / kE
                // terms): what remains -- cost, J' Sigma J, the "current" curvature -- is positive definite by construction; IPOPT's
                // delta_w schedule only if rounding makes even that fail.  (kS / kE are rebuilt by the next assemble_newton.)
                convex = 1;
                for (int k = lane; k < N; k += WAVE) { LD(L::kS + 2 * k) = 0.0; LD(L::kE + 2 * k) = 0.0; }
                SYNC();
                continue;
            }
            dw = tries == 0 ? (dw_last == 0.0 ? 1e-4 : fmax(1e-20, dw_last / 3.0)) : dw * (dw_last == 0.0 ? 100.0 : 8.0);
            tries++;
            if (dw > 1e40) break;
        }
        if (!ok) break;
        if (dw != 0.0) dw_last = dw;
        cvx_run = used_convex ? cvx_run + 1 : 0;
        long long tc5 = CLK();
        riccati_forward<NOBS, NMAX, SweepUnroll<NFIX>::v>(sm, c);
        long long tc6 = CLK();
        // ---- row steps, step lengths, merit pieces ---------------------------------------------------
        // fraction-to-the-boundary without per-row divisions: a = min(1, tau / max_j(-d_j / v_j))
        double rp_max = 0.0, rd_max = 0.0, theta = 0.0, Dphi = 0.0;
        // (jd = J dz of the row; shared tail of the simple-row passes and of the CBF pass)
        auto row_step = [&](int j, bool cnt, bool store, double sc, double jd) {
            const bool on = sc != 0.0;
            const double t = LD(L::rt + j), nu = LD(L::rnu + j), rti = LD(L::rtt + j), rcj = LD(L::rc + j);
            double rwj, rsj;
            row_sig_w<L>(sm, j, on, mu, t, nu, rti, rcj, rsj, rwj);
            ROW_LOADS_DONE();
            const double rp = rcj - t;
            const double dt = sel(on, jd + rp, 0.0);
            // dnu = (mu - t nu - nu dt)/t = mu/t - nu - Sigma dt = -w + Sigma (rp - dt)
            const double dnu = sel(on, -rwj + rsj * (rp - dt), 0.0);
            LD(SINK(store, L::rdt + j)) = dt;
            const double dtr = dt * rti;                       // dt / t
            rp_max = fmax(rp_max, sel(cnt, -dtr, 0.0));
            rd_max = fmax(rd_max, sel(cnt && on, -dnu * frcp(nu), 0.0));
            theta += sel(cnt && on, fabs(rp), 0.0);
            Dphi -= sel(cnt && on, mu * dtr, 0.0);
        };
        ROWS(j, jv, lane, m) {
            // J dz straight from the step (differencing row values would lose eps*|x|, which the
            // multiplier update amplifies by Sigma = nu/t ~ 1e10..1e13)
            const bool simple = !ROW_IS_CBF(j, N);
            const int pk = RIVT(si, j);
            const double sc = row_scale<L>(sm, si, j, N), jd = RIV_SGN(pk) * LD(L::dZ + RIV_IDX(pk));
            row_step(j, jv && simple, simple, sc, jd);
        }
        if (NOBS) {
            CBF_ROWS(j, k, ob, ev, lane, N) {   // the full Jacobian row
                const double* J = sm + L::Jc + (k * L::NO + ob) * NZ;
                double jc = 0.0;
#pragma unroll
                for (int a = 0; a < NZ; a++) jc += J[a] * LD(L::dZ + k * NZ + a);
                row_step(j, ev, true, cbf_scale<L>(sm, k, ob), jc);
            }
        }
        wave_max2(rp_max, rd_max);
        const double a_p = (rp_max > tau) ? tau / rp_max : 1.0;
        const double a_d = (rd_max > tau) ? tau / rd_max : 1.0;
        double cost_qq;
        double cost_d = cost_dir<NOBS, NMAX>(sm, c, cost_qq);       // lane partials
        wave_sum4(theta, Dphi, cost_d, cost_qq);                    // four sums, one row reduction (crx_wave.h)
        Dphi += cost_d;
        const double phi0 = f - mu * logsum_t;
        if (theta_min < 0.0) {
            theta_min = 1e-4 * fmax(1.0, theta);
            theta_max = 1e4 * fmax(1.0, theta);
        }
        long long tc7 = CLK();
        // ---- filter line search ----------------------------------------------------------------------
        double al = a_p, fn = f, lt_acc = logsum_t;
        int acc = 0, ftype = 0;
        // switching condition al * (-Dphi)^2.3 > theta^1.1 (only consulted when theta <= theta_min)
        const bool sw_try = (theta <= theta_min) && (Dphi < 0.0);
        // ... compared in the log2 domain: log2(al) + 2.3 log2(-Dphi) > 1.1 log2(theta), with log2_fast (double exponent +
        // v_log_f32 of the mantissa, ~1e-7 absolute).  Two pow() calls were ~2.6 k cycles of this iteration and kept ~50
        // VGPRs of polynomial constants alive; the test is a heuristic threshold, a tie within 1e-7 may fall either way.
        const double sw_gap = sw_try ? 2.3 * log2_fast(-Dphi) - 1.1 * log2_fast(theta) : 0.0;
        // The backtracking stops at alpha_min = 1e-10 ("no acceptable step"; IPOPT's alpha_min plays the same role): below it
        // a step changes nothing in double precision relative to the iterate, the trial values differ from the current
        // ones by rounding only, and whether the filter happens to accept one of them is noise -- on infeasible problems
        // (slacks collapsed to ~1e-20) that noise used to decide at which iteration the solve gave up.
        for (int ls = 0; ls < 40 && al >= 1e-10; ls++) {
            fn = f + al * (cost_d + al * cost_qq);   // exact: the cost is quadratic along the step
            double thn = 0.0;
            LogAcc lg;
            auto row_trial = [&](int j, bool cnt, bool store, double sc, double cn, double t, double dt) {
                double tn = t + al * dt;
                tn = sel(cn > tn, cn, tn);                     // slack reset
                const bool off = sc == 0.0;
                tn = sel(off, 1.0, tn);
                cn = sel(off, 1.0, cn);
                LD(SINK(store, L::rtt + j)) = tn;
                lg.mul(sel(cnt, tn, 1.0));
                thn += sel(cnt, fabs(cn - tn), 0.0);
            };
            ROWS(j, jv, lane, m) {
                const bool simple = !ROW_IS_CBF(j, N);
                const double sc = row_scale<L>(sm, si, j, N), t = LD(L::rt + j), dt = LD(L::rdt + j), cj = LD(L::rc + j);
                ROW_LOADS_DONE();
                row_trial(j, jv && simple, simple, sc, cj + al * (dt - (cj - t)), t, dt);   // linear rows: exact
            }
            if (NOBS) {
                CBF_ROWS(j, k, ob, ev, lane, N) {   // evaluated at Z + al dZ
                    const double sc = cbf_scale<L>(sm, k, ob), t = LD(L::rt + j), dt = LD(L::rdt + j);
                    row_trial(j, ev, true, sc, sc * cbf_value<NOBS, NMAX>(sm, c, k, ob, al), t, dt);
                }
            }
            lt_acc = lg.wave_total_with(thn);                         // thn and the exponent sum share one reduction
            const double phin = fn - mu * lt_acc;
            int okf = (thn <= theta_max) && (phin == phin);
            {   // filter (nf <= MAXF < WAVE entries: one pass, lanes past nf read entry 0 and are masked)
                const bool iv = lane < nf;
                const int i = iv ? lane : 0;
                const double fth = LD(L::Fth + i), fph = LD(L::Fph + i);
                if (__any(iv && !(thn < fth || phin < fph))) okf = 0;
            }
            if (okf) {
                if (sw_try && log2_fast(al) + sw_gap > 0.0) {
                    if (phin <= phi0 + eta * al * Dphi + 10.0 * 2.2e-16 * fabs(phi0)) { acc = 1; ftype = 1; }
                } else if (thn <= (1.0 - 1e-5) * theta || phin <= phi0 - 1e-8 * theta) {
                    acc = 1;
                }
            }
            if (acc) break;
            al *= 0.5;
        }
        long long tc8 = CLK();
        tph[0] = tc1 - tc0; tph[1] = tc2 - tc1; tph[2] = tc3 - tc2; tph[3] = tc4 - tc3; tph[4] = tc5 - tc4;
        tph[5] = tc6 - tc5; tph[6] = tc7 - tc6; tph[7] = tc8 - tc7;
        if (kp.trace_rows < 0) { tph[0] = tsub[0]; tph[1] = tsub[1]; tph[2] = tsub[2]; tph[3] = tsub[3]; }
        if (kp.trace && b == kp.trace_problem && it < (kp.trace_rows < 0 ? -kp.trace_rows : kp.trace_rows) && lane == 0) {
            double* tr = kp.trace + (size_t)it * 16;
            for (int q = 0; q < 8; q++) tr[8 + q] = (double)(tph[q]);
            tr[0] = e_d; tr[1] = e_p; tr[2] = e_c; tr[3] = mu; tr[4] = al; tr[5] = a_d; tr[6] = dw; tr[7] = acc ? (ftype ? 2.0 : 1.0) : 0.0;
        }
        if (acc && !ftype && nf < MAXF) {
            if (lane == 0) { LD(L::Fth + nf) = (1.0 - 1e-5) * theta; LD(L::Fph + nf) = phi0 - 1e-8 * theta; }
            nf++;
        }
        if (!acc) { ls_failed = 1; break; }
        if (NOBS) {
            jam = (jam_on && al < JAM_ALPHA && e_p > o.tol) ? jam + 1 : 0;
            // stall: STALL_ITERS iterations without a restoration and still infeasible (the same crawl with steps just above JAM_ALPHA).
            // [r5] At 50 -- calibrated on configs[1]: p99 16 iterations, max 30 -- the rule stopped healthy three-obstacle N = 20 solves that
            // converge by themselves after 53..86 iterations: 88 of the 135 non-converged problems of the benched configs[3] batch
            // (tests/golden/cfg4_stopped.npz)
            if (jam_on && n_restore == 0 && it >= STALL_ITERS && e_p > 1e-6) jam = JAM_COUNT;
            if (jam >= JAM_COUNT && n_restore < 2) { ls_failed = 2; break; }
        }
        // ---- accept ------------------------------------------------------------------------------------
        SYNC();
        COORDS(e, ev, lane, N * NZ + NX) {
            const double zn = LD(L::Z + e) + al * LD(L::dZ + e);
            LD(SINK(ev, L::Z + e)) = zn;                // read-modify-write
        }
        SYNC();
        f = fn;
        logsum_t = lt_acc;                           // the trial slacks rtt become the slacks rt below
        // one pass over the rows: multiplier update (from the pre-step row state), new slack, row value at the new
        // iterate (simple rows exactly from Z, CBF rows evaluated), and the two divergence-test reductions
        double numax = 0.0, th = 0.0;
        nus = 0.0; cmax = 0.0; cmin = INFINITY;
        auto row_accept = [&](int j, bool own, double sc, double v) {   // own: this lane is the one that updates row j
            const bool on = sc != 0.0, cnt = own && on;
            const double tn = LD(L::rtt + j), rcj = LD(L::rc + j), rtj = LD(L::rt + j);
            const double rdj = LD(L::rdt + j), rnj = LD(L::rnu + j);
            double rwj, rsj;   // (slim: 1/t is recomputed -- rtt holds the trial slack by now; frcp(t) returns what assemble_newton stored)
            row_sig_w<L>(sm, j, on, mu, rtj, rnj, L::SLIM ? frcp(rtj) : 0.0, rcj, rsj, rwj);
            ROW_LOADS_DONE();
            const double mut = mu * frcp(tn);
            const double rp = rcj - rtj;                       // dnu as in the row-step pass (rc, rt, rw, rsig still hold that state)
            const double dnu = -rwj + rsj * (rp - rdj);
            double nn = rnj + a_d * dnu;
            nn = fmin(fmax(nn, mut * (1.0 / kappa_sigma)), kappa_sigma * mut);
            LD(SINK(cnt, L::rt + j)) = tn;
            LD(SINK(cnt, L::rnu + j)) = nn;             // read-modify-write
            numax = fmax(numax, sel(cnt, nn, 0.0));     // cnt, not on: a lane past the last row re-reads row 0 AFTER its update
            th = fmax(th, sel(cnt, fabs(v - tn), 0.0));
            nus += sel(cnt, nn, 0.0);
            cmax = fmax(cmax, sel(cnt, tn * nn, 0.0));
            cmin = fmin(cmin, sel(cnt, tn * nn, INFINITY));
            LD(SINK(own, L::rc + j)) = sel(on, v, 1.0);
        };
        ROWS(j, jv, lane, m) {
            const int pk = RIVT(si, j);
            const double sc = row_scale<L>(sm, si, j, N), v = RIV_SGN(pk) * (LD(L::Z + RIV_IDX(pk)) - row_bound<L>(sm, c, j));
            row_accept(j, jv && !ROW_IS_CBF(j, N), sc, v);
        }
        if (NOBS) {
            CBF_ROWS(j, k, ob, ev, lane, N) {
                const double sc = cbf_scale<L>(sm, k, ob);
                row_accept(j, ev, sc, sc * cbf_value<NOBS, NMAX>(sm, c, k, ob, 0.0));
            }
        }
        SYNC();
        {   // four extremes, one row reduction: min = -max(-.)
            double ncmin = -cmin;
            wave_max4(numax, th, cmax, ncmin);
            cmin = -ncmin;
        }
        nus = wave_sum(nus);
        e_p = th;
        first_order<NOBS, NMAX>(sm, si, c);
        if (kp.trace && b == kp.trace_problem && it < (kp.trace_rows < 0 ? -kp.trace_rows : kp.trace_rows) && lane == 0 && kp.trace_rows > 0)
            kp.trace[(size_t)it * 16 + 8] = (double)(tph[0] + (CLK() - tc8));   // slot 8: KKT rows + accept/first-order
        if (numax > 1e12 && th > 1e-6) { status = CRX_STALLED; it++; break; }   // IPOPT's divergence heuristic: not a proof
        // still violated after the step: look for the proof that it must be (linear rows only)
        if (NOBS == 0 && th > 1e-6) {
            if (box_certificate<NOBS, NMAX>(sm, si, c, kp.delta_max, kp.a_max) < -1e-8 * numax) { status = 2; it++; break; }
        }
    }
    if (!ls_failed) break;
    // [r4] crash path (ii): the solve started at the reference's zero point and stalls on violated CBF rows -- restart ONCE from the
    // feasible interior point; later failures, or no candidate, fall through to the closed-form restoration
    int full = 0;
    if (NOBS && crash_path && n_restore == 0 && !crash && crash_cand >= 0) {
        crash_write<NOBS, NMAX, false>(sm, c, kp, crash_cand);   // the candidate's inputs and states, sigma = push: restore_slacks() below raises the cascade
        crash = 1; full = 1;
    }
    if (NOBS && o.restore_iters >= 0 && n_restore < 2 && (__builtin_amdgcn_readfirstlane((int)restore_slacks<NOBS, NMAX>(sm, c, o.slack_push)) | full)) {
        // slacks of the CBF rows and of the sigma bounds re-initialised like at the start, multipliers centred; after a crash restart
        // (full) the inputs have changed as well: every other row restarts with t = max(|c|, push), nu = 1
        eval_rows<NOBS, NMAX>(sm, si, c);
        SYNC();
        for (int j = lane; j < m; j += WAVE) {
            const bool cbf = ROW_IS_CBF(j, N);
            const int r = j < N * NR ? j % NR : 8;              // rows N*NR.. are the sigma_0 bounds
            const bool sig = NOBS && (j >= N * NR || (r >= 8 && r < 8 + NOBS));
            if (row_scale<L>(sm, si, j, N) != 0.0) {
                const double t = fmax(fabs(LD(L::rc + j)), o.slack_push);
                if (cbf || sig) {
                    LD(L::rt + j) = t;
                    LD(L::rnu + j) = fmin(fmax(o.mu_init / t, 1e-8), 1e8);
                } else if (full) {
                    LD(L::rt + j) = t;
                    LD(L::rnu + j) = 1.0;
                }
            }
        }
        SYNC();
        first_order<NOBS, NMAX>(sm, si, c);
        f = cost_value<NOBS, NMAX>(sm, c, 0.0);
        row_stats();
        if (n_restore++ == 0) it_limit = it + 1 + o.restore_iters;
        mu = o.mu_init; nf = 0; theta_min = -1.0; theta_max = INFINITY; dw_last = 0.0; status = 1; jam = 0;
        it++;                                            // the abandoned step was an iteration (oracle: `continue`)
        continue;
    }
    if (ls_failed == 2) {
        // jammed, but nothing to restore (the violated rows are not CBF rows): stop looking for jams and redo this
        // iteration -- same state, same step, accepted this time (the oracle simply goes on to accept it)
        jam_on = 0; jam = 0;
        // (slim layout: the curvature table G shares its storage with the feedback gains the abandoned sweep has just written)
        if constexpr (L::SLIM) first_order<NOBS, NMAX>(sm, si, c);
        continue;
    }
    // no acceptable step and nothing to restore: IPOPT's "converged to a point of local infeasibility" / "restoration failed" if the
    // constraints are still violated there -- not a proof: CRX_STALLED
    if (e_p > 1e-6) status = CRX_STALLED;
    break;
    }
    }   // QPM
    if constexpr (SPEC) {   // the solve is over: release the speculating wave
        if (lane == 0) LD(L::ctl) = 0.0;
        __syncthreads();
    }
    if (infeas0) status = CRX_INFEASIBLE;        // a bound violated by the fixed x_0: proved

    // ---- write back: one coalesced pass ----------------------------------------------------------------
    SYNC();
    // [r4b] the lane index is taken afresh here: the addresses of this pass are formed now instead of being carried (in AGPR slots, in the
    // general instantiations) from the top of the kernel across the whole solve -- DESIGN.md section 8, second observation
    int lane_wb = threadIdx.x;
    asm volatile("" : "+v"(lane_wb));
    {
    const int lane = lane_wb;   // shadows the kernel's `lane` inside this block
    double* Xb = kp.X + (size_t)b * (N + 1) * 6;
    double* Ub = kp.U + (size_t)b * N * 2;
    if (kp.mode == 0 && status != 0) {
        planner_fallback<L>(sm, kp, b, lane, N);
    } else {
        for (int e = lane; e < (N + 1) * 6; e += WAVE) { const int k = e / 6, i = e - k * 6; Xb[e] = LD(L::Z + k * NZ + i); }
        for (int e = lane; e < N * 2; e += WAVE) Ub[e] = LD(L::Z + (e >> 1) * NZ + NX + (e & 1));
        if (lane == 0) kp.cost[b] = f;
    }
    if (kp.mode == 1 && kp.sigma) {
        for (int e = lane; e < kp.n_obs_max * (N + 1); e += WAVE) {
            const int ob = e / (N + 1), k = e - ob * (N + 1);
            kp.sigma[(size_t)b * kp.n_obs_max * (N + 1) + e] = (NOBS && ob < c.nobs) ? LD(L::Z + k * NZ + 6 + ob) : 0.0;
        }
    }
    double Eout = E0;
    if (CRX_KKT_DIAG && kp.kkt_unscaled && status == 0) {
        // diagnostics (crx_debug_kkt_unscaled): the converged exit is taken right after dual_infeasibility(), so ga holds the reduced Lagrangian gradient
        // of the returned iterate; the row arrays hold its slacks, multipliers and (scaled) row values.  Outside every loop: no register is carried for it.
        // kkt_unscaled = 1: the max of the three, 2: dual infeasibility, 3: constraint violation, 4: complementarity
        double ed = 0.0, ev = 0.0, ec = 0.0;
        for (int e = lane; e < N * NZ + NX; e += WAVE) ed = fmax(ed, fabs(LD(L::ga + e)));
        for (int j = lane; j < m; j += WAVE) {
            const double sc = row_scale<L>(sm, si, j, N);
            if (sc != 0.0) { ev = fmax(ev, fabs(LD(L::rc + j) - LD(L::rt + j)) / sc); ec = fmax(ec, LD(L::rt + j) * LD(L::rnu + j)); }
        }
        const int km = kp.kkt_unscaled;
        Eout = wave_max(km == 2 ? ed : km == 3 ? ev : km == 4 ? ec : fmax(ed, fmax(ev, ec)));
    }
    if (lane == 0) { kp.status[b] = status; kp.kkt[b] = Eout; kp.iters[b] = it; }
    }
}

// This file is compiled TWICE (Makefile): as it stands for the planner instantiations <0, *> and everything else in it, and
// through crx_kernels_obs.hip (CRX_TU_OBSTACLES) for the obstacle instantiations <1..3, *> alone -- with the machine scheduler's
// iterative-ilp strategy, which is worth +3.7 % on BASELINE configs[1] (0.991 -> 0.956 ms per 256 NLPs) and +7 % on configs[3] to
// the obstacle kernels and costs the planner kernels 1 % (tools/gpu_round3_l.sh: max-ilp, max-memory-clause, iterative-minreg
// and the occupancy / latency bias were measured beside it).  Same arithmetic either way: the schedule does not re-associate.
#ifndef CRX_TU_OBSTACLES
// ------------------------------------------------------------------------------------------------
// (6) region selection (planning/overtake_traj_planner.py:205-246): one wave per scenario, lanes
// over (side, stage) collision tests.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) crx_select_kernel(const crx_select_kparams sp) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= sp.n_scen) return;
    const int N = sp.N, V = sp.V, R = V + 1;
    const int nv = min(max(sp.n_veh[s], 0), V);   // the *_dev entry points cannot validate device-resident counts: clamp
    const double r2 = sp.veh_length * sp.veh_length + sp.veh_width * sp.veh_width;
    double best_c = INFINITY;
    int best = 0;
    for (int r = 0; r < R; r++) {
        double cst = INFINITY;
        if (r <= nv) {
            const double* Xr = sp.X + (((size_t)s * R + r) * (N + 1)) * 6;
            double hits = 0.0;
            for (int e = lane; e < 2 * (N + 1); e += WAVE) {
                const int side = e / (N + 1), j = e - side * (N + 1);
                const int v = side == 0 ? r - 1 : r;                                          // :213, :227
                if (v < 0 || v >= nv) continue;
                double os = sp.obs_s[((size_t)s * V + v) * (N + 1) + j];
                os = wrap_above(os, sp.lap_length);                                             // :216-217
                const double ds = Xr[6 * j + 4] - os;
                const double de = Xr[6 * j + 5] - sp.obs_ey[((size_t)s * V + v) * (N + 1) + j];
                if (!(ds * ds + de * de - r2 >= 0.0)) hits += 1.0;                            // :220-223
            }
            hits = wave_sum(hits);
            cst = -sp.w_prog * (Xr[6 * N + 4] - Xr[4]) + sp.w_coll * hits;                    // :209
            if (sp.old_flag[s] >= 0 && sp.old_flag[s] != r) cst += sp.w_switch;               // :238-243
            if (cst < best_c) { best_c = cst; best = r; }                                     // first arg-min :244
        }
        if (lane == 0) sp.sel_cost[(size_t)s * R + r] = cst;
    }
    if (lane == 0) sp.flag[s] = best;
    const double* Xb = sp.X + (((size_t)s * R + best) * (N + 1)) * 6;
    for (int e = lane; e < (N + 1) * 6; e += WAVE) sp.best_X[(size_t)s * (N + 1) * 6 + e] = Xb[e];
}

#endif  // !CRX_TU_OBSTACLES

#ifdef CRX_PROBE_ONE
// tools/kernel_resources.py one NOBS NMAX DEG NFIX: ONE instantiation, compiled alone (seconds instead of minutes)
template __global__ void crx_solve_kernel<CRX_PROBE_ONE>(const crx_kparams);
#else
// ------------------------------------------------------------------------------------------------
// (7) launchers (plain C++ linkage inside the library; the C ABI lives in crx_api.hip)
// ------------------------------------------------------------------------------------------------
template <int NOBS, int NMAX, int DEG = 0, int NFIX = 0, int SPEC = 0, int QPM = 0>
static hipError_t launch_t(const crx_kparams& kp, hipStream_t st) {
    if constexpr (NOBS == 0 && QPM == 0) {   // the all-linear problems: crx_ipm_opts.qp_method picks the instantiation (0 = predictor-corrector, 1 = filter line search)
        if (kp.opts.qp_method != 0) return launch_t<NOBS, NMAX, DEG, NFIX, SPEC, 1>(kp, st);
    }
#if CRX_STATIC_LDS
    hipLaunchKernelGGL((crx_solve_kernel<NOBS, NMAX, DEG, NFIX, SPEC, QPM>), dim3(kp.batch), dim3(WAVE * (1 + SPEC)), 0, st, kp);   // the layout is a static array of the kernel
#else
    static_assert(SPEC == 0, "the two-wave instantiations are static-LDS kernels");
    const size_t bytes = Lay<NOBS, NMAX>::BYTES;
    // the opt-in to > 64 KiB of dynamic LDS is a property of the (function, device) pair: set once per device, not per launch
    static int attr_set_on = -1;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (attr_set_on != dev) {
        hipError_t e = hipFuncSetAttribute((const void*)crx_solve_kernel<NOBS, NMAX, DEG, NFIX, 0, QPM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        attr_set_on = dev;
    }
    hipLaunchKernelGGL((crx_solve_kernel<NOBS, NMAX, DEG, NFIX, 0, QPM>), dim3(kp.batch), dim3(WAVE), bytes, st, kp);
#endif
    return hipGetLastError();
}
#ifdef CRX_TU_SPEC
// [r6] crx_kernels_spec.hip: the two-wave (speculating) instantiations of the one-obstacle, degree-6 kernel at the tuned horizons 12 and 10 --
// BASELINE configs[1] and the reference's default horizon -- and nothing else.  crx_api.hip routes a launch here when it leaves SIMDs idle.
hipError_t crx_launch_solve_spec(const crx_kparams& kp, hipStream_t st) {
    if (kp.batch == 0) return hipSuccess;
    if (kp.degree != 6) return hipErrorInvalidValue;
    if (kp.N == 12) return launch_t<1, 12, 6, 12, 1>(kp, st);
    if (kp.N == 10) return launch_t<1, 12, 6, 10, 1>(kp, st);
    return hipErrorInvalidValue;
}
#else
// [r5] The GENERAL instantiations (run-time horizon NFIX = 0, run-time exponent DEG = 0, the generic 4..6-obstacle ones) live in a third
// translation unit, crx_kernels_gen.hip, built conservatively: 256 registers and no AGPRs, no inline-assembly DPP,
// nothing lane-derived hoisted out of the interior-point loop.  This unit and the obstacle unit only hold the tuned fixed-horizon ones.
hipError_t crx_launch_solve_general(const crx_kparams& kp, int nobs_template, hipStream_t st);
int crx_solve_resident_per_cu_general(int N, int nobs_template);
#if !CRX_TU_GENERAL
// fixed-horizon (tuned) instantiations [r6: a BUILD PARAMETER].  CRX_NFIX_LIST -- `make NFIX_LIST=10,12,16,20` -- names the horizons that get an
// instantiation with the horizon as a compile-time constant (straight-line sweeps, immediates for every stage address: 1.5 .. 2x the general unit's
// speed, profiles/r06_variants.txt) for every obstacle count 0 .. 3; any other horizon runs on the general unit.  Default: 10 (the reference's
// defaults: utils/base.py:281, :390), 12 (BASELINE configs[1], [2], [4]), 20 (configs[3]).  Layout class of a horizon: <= 12 -> 12; <= 20 with three
// obstacle slots -> 20 (the slim layout, four problems per CU); otherwise CRX_MAX_N.  Budget per entry: one kernel per obstacle count (two for the
// planner: crx_ipm_opts.qp_method), ~100 KB of code and 15 .. 40 s of build time each; registers / LDS as its layout class.
#ifndef CRX_NFIX_LIST
#define CRX_NFIX_LIST 10, 12, 20
#endif
template <int NOBS, int NF> struct NfixLayout { static constexpr int v = NF <= 12 ? 12 : ((NOBS == 3 && NF <= 20) ? 20 : CRX_MAX_N); };
template <int NOBS, int DEG, int... NFS>
static bool launch_fixed(const crx_kparams& kp, hipStream_t st, hipError_t& e) {
    bool hit = false;
    (void)((kp.N == NFS ? (e = launch_t<NOBS, NfixLayout<NOBS, NFS>::v, DEG, NFS>(kp, st), hit = true) : false) || ...);
    return hit;
}
// obstacle instantiations: the degree-6 one for the reference's exponent; other exponents (2 / 4 / 8) take the general unit
template <int NOBS>
static hipError_t launch_n(const crx_kparams& kp, hipStream_t st) {
    hipError_t e = hipSuccess;
    (void)e;
#if CRX_NFIX
    if constexpr (NOBS > 0 && CRX_DEG6) {
        if (kp.degree == 6 && launch_fixed<NOBS, 6, CRX_NFIX_LIST>(kp, st, e)) return e;
    } else {
        if (launch_fixed<NOBS, 0, CRX_NFIX_LIST>(kp, st, e)) return e;
    }
#endif
    return crx_launch_solve_general(kp, NOBS, st);
}
#endif  // !CRX_TU_GENERAL

// the obstacle instantiations live in the other translation unit
hipError_t crx_launch_solve_obs(const crx_kparams& kp, int nobs_template, hipStream_t st);
int crx_solve_resident_per_cu_obs(int N, int nobs_template);

#if CRX_TU_GENERAL
// horizon class of a general launch: the smallest layout that holds kp.N
template <int NOBS, int DEG>
static hipError_t launch_g(const crx_kparams& kp, hipStream_t st) {
    if (kp.N <= 12) return launch_t<NOBS, 12, DEG, 0>(kp, st);
    if constexpr (NOBS == 3) {
        if (kp.N <= 20) return launch_t<3, 20, DEG, 0>(kp, st);
    }
    return launch_t<NOBS, CRX_MAX_N, DEG, 0>(kp, st);
}
template <int NOBS>
static hipError_t launch_gd(const crx_kparams& kp, hipStream_t st) {
    if constexpr (NOBS > 0 && CRX_DEG6) {
        if (kp.degree == 6) return launch_g<NOBS, 6>(kp, st);
    }
    return launch_g<NOBS, 0>(kp, st);
}
hipError_t crx_launch_solve_general(const crx_kparams& kp, int nobs_template, hipStream_t st) {
    switch (nobs_template) {
        case 0: return launch_gd<0>(kp, st);
        case 1: return launch_gd<1>(kp, st);
        case 2: return launch_gd<2>(kp, st);
        case 3: return launch_gd<3>(kp, st);
        // [r4] four to six obstacles (CRX_MAX_OBS = 6: the reference admits any number, control.py:524-562): ONE generic instantiation
        // per horizon class -- exponent and horizon read at run time, the Riccati update in two passes, the sigma columns of T in
        // three; the slow path: correct first
        case 4: case 5: case 6: return kp.N <= 12 ? launch_t<6, 12, 0, 0>(kp, st) : launch_t<6, CRX_MAX_N, 0, 0>(kp, st);
        default: return hipErrorInvalidValue;
    }
}
#elif defined(CRX_TU_OBSTACLES)
hipError_t crx_launch_solve_obs(const crx_kparams& kp, int nobs_template, hipStream_t st) {
    switch (nobs_template) {
        case 1: return launch_n<1>(kp, st);
        case 2: return launch_n<2>(kp, st);
        case 3: return launch_n<3>(kp, st);
        case 4: case 5: case 6: return crx_launch_solve_general(kp, nobs_template, st);
        default: return hipErrorInvalidValue;
    }
}
#else
hipError_t crx_launch_solve(const crx_kparams& kp, int nobs_template, hipStream_t st) {
    if (kp.batch == 0) return hipSuccess;
    if (nobs_template == 0) return launch_n<0>(kp, st);
    return crx_launch_solve_obs(kp, nobs_template, st);
}

size_t crx_solve_lds_bytes(int N, int nobs_template) {
    const bool small = N <= 12;
    switch (nobs_template) {
        case 0: return small ? Lay<0, 12>::BYTES : Lay<0, CRX_MAX_N>::BYTES;
        case 1: return small ? Lay<1, 12>::BYTES : Lay<1, CRX_MAX_N>::BYTES;
        case 2: return small ? Lay<2, 12>::BYTES : Lay<2, CRX_MAX_N>::BYTES;
        case 3: return small ? Lay<3, 12>::BYTES : (N <= 20 ? Lay<3, 20>::BYTES : Lay<3, CRX_MAX_N>::BYTES);
        default: return small ? Lay<6, 12>::BYTES : Lay<6, CRX_MAX_N>::BYTES;
    }
}

#endif  // translation unit

// resident single-wave workgroups per CU of the instantiation that WOULD RUN (N, nobs_template) with the reference's exponent: the
// runtime's answer, i.e. min over the LDS and the register file, for the same (DEG, NFIX) selection as launch_d / launch_h -- the
// fixed-horizon instantiations are other kernels than the general ones, with their own register counts
template <int NOBS, int NMAX, int NFIX>
static int occ_t() {
    int n = 0;
    constexpr int DEG = (NOBS > 0 && NOBS <= 3 && CRX_DEG6) ? 6 : 0;   // (the generic six-obstacle instantiation reads the exponent at run time)
#if CRX_STATIC_LDS
    const size_t bytes = 0;   // the layout is static LDS of the kernel: the runtime counts it by itself
#else
    const size_t bytes = Lay<NOBS, NMAX>::BYTES;
    if (hipFuncSetAttribute((const void*)crx_solve_kernel<NOBS, NMAX, DEG, NFIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return -1;
#endif
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, crx_solve_kernel<NOBS, NMAX, DEG, NFIX>, WAVE, bytes) != hipSuccess) return -1;
    return n;
}
#if CRX_TU_GENERAL
template <int NOBS>
static int occ_g(int N) {
    if (N <= 12) return occ_t<NOBS, 12, 0>();
    if constexpr (NOBS == 3) {
        if (N <= 20) return occ_t<3, 20, 0>();
    }
    return occ_t<NOBS, CRX_MAX_N, 0>();
}
int crx_solve_resident_per_cu_general(int N, int nobs_template) {
    switch (nobs_template) {
        case 0: return occ_g<0>(N);
        case 1: return occ_g<1>(N);
        case 2: return occ_g<2>(N);
        case 3: return occ_g<3>(N);
        default: return N <= 12 ? occ_t<6, 12, 0>() : occ_t<6, CRX_MAX_N, 0>();
    }
}
#else
template <int NOBS, int... NFS>
static bool occ_fixed(int N, int& n) {
    bool hit = false;
    (void)((N == NFS ? (n = occ_t<NOBS, NfixLayout<NOBS, NFS>::v, NFS>(), hit = true) : false) || ...);
    return hit;
}
template <int NOBS>
static int occ_n(int N) {
#if CRX_NFIX
    int n = 0;
    if (occ_fixed<NOBS, CRX_NFIX_LIST>(N, n)) return n;
#endif
    return crx_solve_resident_per_cu_general(N, NOBS);
}
#endif
#if CRX_TU_GENERAL
#elif defined(CRX_TU_OBSTACLES)
int crx_solve_resident_per_cu_obs(int N, int nobs_template) {
    switch (nobs_template) {
        case 1: return occ_n<1>(N);
        case 2: return occ_n<2>(N);
        case 3: return occ_n<3>(N);
        default: return crx_solve_resident_per_cu_general(N, nobs_template);
    }
}
#else
int crx_solve_resident_per_cu(int N, int nobs_template) {
    return nobs_template == 0 ? occ_n<0>(N) : crx_solve_resident_per_cu_obs(N, nobs_template);
}

hipError_t crx_launch_select(const crx_select_kparams& sp, hipStream_t st) {
    if (sp.n_scen == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_select_kernel, dim3(sp.n_scen), dim3(WAVE), 0, st, sp);
    return hipGetLastError();
}

// diagnostics (crx_debug_wave_reduce, not in crx.h): the packed wave reductions of crx_wave.h on four 64-lane inputs.
// out [16 + 3 * 64]: out[0..3] wave_sum4, [4..7] wave_max4, [8..9] wave_sum2 of inputs 0 and 1, [10..11] wave_max2 of inputs 2 and 3,
// [12..15] the single-value reductions (sum of 0, max of 1, min of 2, product of the mantissas of 3).
__global__ void __launch_bounds__(WAVE) crx_debug_reduce_kernel(const double* in, double* out) {
    const int lane = threadIdx.x;
    double a = in[lane], b = in[64 + lane], c = in[128 + lane], d = in[192 + lane];
    double s0 = a, s1 = b, s2 = c, s3 = d;
    wave_sum4(s0, s1, s2, s3);
    double m0 = a, m1 = b, m2 = c, m3 = d;
    wave_max4(m0, m1, m2, m3);
    double p0 = a, p1 = b;
    wave_sum2(p0, p1);
    double q0 = c, q1 = d;
    wave_max2(q0, q1);
    int ex;
    const double r0 = wave_sum(a), r1 = wave_max(b), r2 = wave_min(c), r3 = wave_prod(frexp(fabs(d) + 1.0, &ex));
    if (lane == 0) {
        out[0] = s0; out[1] = s1; out[2] = s2; out[3] = s3; out[4] = m0; out[5] = m1; out[6] = m2; out[7] = m3;
        out[8] = p0; out[9] = p1; out[10] = q0; out[11] = q1; out[12] = r0; out[13] = r1; out[14] = r2; out[15] = r3;
    }
    // [r4] row_dot (v_fmac_f64_dpp row_newbcast): out[16 + lane] = c + sum_{i<7} m_i * (lane i of a's 16-lane row), m_i = (i + 1) b + d, under
    // full EXEC; out[80 + lane] = the same with terms 3 .. 9 inside an `if (lane < 10)` region (the sweeps' predication; NaN elsewhere);
    // out[144 + lane] = a three-term dot product of lanes 7 .. 9 whose x is the value just accumulated (the forward sweep's input term)
    double m[7];
#pragma unroll
    for (int i = 0; i < 7; i++) m[i] = (double)(i + 1) * b + d;
    out[16 + lane] = row_dot<7, 0>(a, m, c);
    double masked = __longlong_as_double(0x7ff8000000000000LL), chained = masked;
    if (lane < 10) {
        masked = row_dot<7, 3>(a, m, c);
        const double first = row_dot<7, 0>(a, m, c);
        chained = row_dot<3, 7>(first, m, first);
    }
    out[80 + lane] = masked;
    out[144 + lane] = chained;
}

hipError_t crx_launch_debug_reduce(const double* in, double* out, hipStream_t st) {
    hipLaunchKernelGGL(crx_debug_reduce_kernel, dim3(1), dim3(WAVE), 0, st, in, out);
    return hipGetLastError();
}
#endif  // !CRX_TU_OBSTACLES
#endif  // !CRX_TU_SPEC
#endif  // !CRX_PROBE_ONE
