// crx_lmpc.hip -- gfx950 kernel for the learning-MPC QP of car-racing (SURVEY.md section 8f row 1;
// reference: /root/reference/car_racing/control/control.py:610-730 `lmpc`).
//
// One QP per 64-lane wavefront, everything in that wave's LDS slice (38.9 KB at N <= 12: four problems per CU), FP64,
// no MFMA (largest dense block 30x30, factorised once per iteration along a serial dependency chain).
//
//   variables   u_0..u_{N-1} (2N), lambd (M <= 60); states eliminated by the affine LTV roll-out
//               x_k = xf_k + S_k u   (S_k = dx_k/du, built once per problem)
//   rows        4N input-box rows, 3(N-1) state rows (vx_k <= v_max, |ey_k| <= w; k = 1..N-1),
//               M rows lambd >= 0
//   equalities  x_N - SS lambd = 0  (6),  1'lambd = 1
//   second attempt (include/crx.h, crx_lmpc_solve): x_0 = xcurv + w with cost w_x0 w'w; the six w's are
//               appended to the u block (columns 2N..2N+5 of S hold dx_k/dx_0)
//
// Interior-point iteration: identical to oracle/crx_oracle_lmpc.c (slack form, monotone barrier,
// fraction-to-the-boundary, filter line search, y updated with the primal step length).  Newton
// system by block elimination, every factor a Cholesky:
//   K_u = H_u + J_u' Sigma J_u  (2N x 2N)          -> L_u, with Phi and rhs_u carried as extra rows,
//                                                      so  Y = L_u^-1 Phi',  z = L_u^-1 rhs_u  come for free
//   W   = Y Y'                     (6 x 6)          -> L_w   (per lane, registers)
//   G   = D_lambda + T T',  T = SS' L_w^-T (M x 6) -> never assembled: product form L_1..L_6 D' L_6'..L_1' by six positive
//                                                      rank-one updates of the diagonal factor, every triangular solve
//                                                      ONE wave prefix / suffix scan (lane j = element j; see the block
//                                                      comment in the kernel)
//   dy_1 from 1'dlambd = -e_1, then dlambd, dy_x, du by back substitution.
// A Schur complement on the 7 equalities (E K^-1 E') is NOT used: the active lambd's carry no
// curvature but the barrier's, so K^-1 spans 1e-16..1e16 near the solution; G stays well scaled.
// Lane i owns row i of L_u (left-looking Cholesky blocked by four columns: rows are read as LDS broadcasts, the pivots of
// a diagonal block are broadcast with v_readlane), 1/sqrt by v_rsq_f64 + 2 Newton steps.  The oracle factorises G with a
// dense Cholesky: the parity tests compare the two routes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "crx_kparams.h"
#include "crx_wave.h"

#define LMAXF 16
#ifndef CRX_LMPC_LATE
#define CRX_LMPC_LATE 25   // stagnation rule: iterations with mu < 1e-6 before the QP is left on its noise floor (oracle: LATE_ITERS)
#endif
#ifndef CRX_STATIC_LDS
#define CRX_STATIC_LDS 1   // 0: dynamic (extern) LDS as up to libcrx 0.2.0 (A/B builds)
#endif
#define CRX_LMPC_SS44 44   // the reference's num_ss_points (utils/base.py:357): capacity of the six-per-CU instantiation

template <int NMAX, bool DENSE = true, int MSS = CRX_MAX_SS>
struct LL {
    static constexpr int NU2 = 2 * NMAX + 6 /* inputs + initial-state relaxation */, LDK = NU2 + 1, KR = NU2 + 7;
    static constexpr int MS = MSS;   // capacity of the safe-set arrays: CRX_MAX_SS, or 44 (the reference's num_ss_points) for the 6-per-CU instantiation
    static constexpr int MR = 4 * NMAX + 3 * (NMAX - 1) + MS;
    // offsets in doubles.  A, B, C are read only while the roll-out xf and the sensitivities S are built (before the first
    // iteration) and share the storage of K, which the first K_u assembly overwrites.
    static constexpr int K = 0;                                // KR x LDK
    static constexpr int A = K, B = A + 36 * NMAX, C = B + 12 * NMAX;
    static_assert(54 * NMAX <= KR * LDK, "A, B, C fit under K");
    static constexpr int xf = K + KR * LDK;
    // sensitivities dx_k/d(u, w), only the rows that are read after set-up: vx and ey of stages 1..N-1 (state rows), all six
    // components of stage N (terminal equality).  Row r of the table is S + r * NU2.
    static constexpr int SROWS = 2 * (NMAX - 1) + 6;
    static constexpr int S = xf + 6 * (NMAX + 1);
    static constexpr int srow(int k, int ey) { return 2 * (k - 1) + ey; }          // 1 <= k <= N-1
    static constexpr int srowN(int c6) { return 2 * (NMAX - 1) + c6; }              // stage N
    static constexpr int Hu = S + SROWS * NU2;                 // NU2 x NU2, row-major, stride NU2
    // DENSE = false: no tracking cost (Q == 0, the reference's default, utils/base.py:353): Hu = 2 R + the dR difference
    // stencil is diagonal + two off-diagonals and is never stored (hu_entry / hu_dot below) -- 7.2 KB less at N = 12,
    // 38 864 -> 31 664 B = five instead of four QPs per CU
    static constexpr int SS = Hu + (DENSE ? NU2 * NU2 : 0);    // [6][MS]
    static constexpr int qf = SS + 6 * MS;
    static constexpr int u = qf + MS, du = u + NU2, g0u = du + NU2, gu = g0u + NU2, ru = gu + NU2;
    static constexpr int lam = ru + NU2, dlam = lam + MS, rl = dlam + MS;
    static constexpr int y = rl + MS, dy = y + 8, e = dy + 8, bx = e + 8, Wt = bx + 8;
    // [r3] 31 664 -> 27 248 B at NMAX = 12, MS = 44: SIX instead of five QPs per CU (<= 27 264 B).  The inverse pivots of L_u live in
    // the padding column of K (ik + j * IKS); rp = c - t and dt = rp + J dz are no longer stored but formed where they are
    // read, by the same operations on the same operands (c stays intact through the iteration: dnu, dead after the K_u
    // assembly, takes the multiplier step that used to be parked in c) -- identical bits
    static constexpr int ik = K + LDK + NU2, IKS = LDK;        // padding column of rows 1..NU2 (row 0's is l_chol's store sink)
    static constexpr int t = Wt + 36, nu = t + MR, c = nu + MR, dnu = c + MR, wv = dnu + MR;
    static constexpr int w0 = wv + MR, w5 = w0 + NMAX;
    static constexpr int Fth = w5 + NMAX, Fph = Fth + LMAXF;
    static constexpr int dmy = Fph + LMAXF;                    // sink of address-predicated stores
    static constexpr int END = dmy + 2;
    static constexpr int TB = t;                               // set-up / write-back scratch over the row arrays: one stage of the
                                                               // full sensitivity block [6][NU2]; the plan X [NMAX+1][6]
    static_assert(6 * NU2 <= 5 * MR && 6 * (NMAX + 1) <= 5 * MR, "scratch fits");
    // (G = D + T T' is never assembled -- product-form factorisation in registers -- so the footprint no longer depends on
    // the safe-set size; with the compact sensitivity table 38.9 KB at NMAX = 12 = four problems per CU, 77 KB / two before)
    static constexpr size_t bytes(int /*n_ss_max*/) { return (size_t)END * 8; }
};

struct LCtx {
    int N, nu2, nv, M, m, el, lane;
    int r_st, r_lam, r_el;
};

// Hu of the Q == 0 case without storing it: row a of the inputs block is [.. -2 dR .. 2 R + 2 dR (+ 2 dR) .. -2 dR ..] at
// columns a - 2, a, a + 2 (control.py:667-681: R u^2 + dR (u_i - u_{i-1})^2), the relaxation block 2 w_x0 on its diagonal.
__device__ __forceinline__ double hu_band_diag(const LCtx& x, const crx_lmpc_kparams& kp, int a) {
    const int i = a >> 1, cc = a & 1;
    const double R = cc ? kp.R[1] : kp.R[0], dR = cc ? kp.dR[1] : kp.dR[0];
    return a < x.nu2 ? 2.0 * R + 2.0 * dR + (i + 1 < x.N ? 2.0 * dR : 0.0) : 2.0 * kp.w_x0;
}
template <class L, bool DENSE>
__device__ __forceinline__ double hu_entry(const double* sm, const LCtx& x, const crx_lmpc_kparams& kp, int a, int b) {
    if (DENSE) return sm[L::Hu + a * L::NU2 + b];
    const double off = (a < x.nu2 && b < x.nu2 && (b == a - 2 || b == a + 2)) ? -2.0 * ((a & 1) ? kp.dR[1] : kp.dR[0]) : 0.0;
    return b == a ? hu_band_diag(x, kp, a) : off;
}
// init + (Hu v)[a] exactly as the dense two-accumulator loop forms it (s0 over the even columns starting at init, s1 over
// the odd ones starting at 0, ascending; the zero entries contribute fma(0, v, s) = s): the three band terms of row a
// share its parity, hence one accumulator.
template <class L, bool DENSE>
__device__ __forceinline__ double hu_dot(const double* sm, const LCtx& x, const crx_lmpc_kparams& kp, int a, int voff, double init) {
    if (DENSE) {
        double s0 = init, s1 = 0.0;
#pragma unroll 4
        for (int b = 0; b < x.nv; b += 2) {
            s0 = fma(sm[L::Hu + a * L::NU2 + b], sm[voff + b], s0);
            s1 = fma(sm[L::Hu + a * L::NU2 + b + 1], sm[voff + b + 1], s1);
        }
        return s0 + s1;
    }
    const bool inp = a < x.nu2, even = (a & 1) == 0;
    const int i = a >> 1;
    const double dR2 = -2.0 * ((a & 1) ? kp.dR[1] : kp.dR[0]);
    const double hm = (inp && i > 0) ? dR2 : 0.0, hp = (inp && i + 1 < x.N) ? dR2 : 0.0, dd = hu_band_diag(x, kp, a);
    const double vm = sm[voff + (a >= 2 ? a - 2 : 0)], v0 = sm[voff + a], vp = sm[voff + (a + 2 < x.nv ? a + 2 : a)];
    double acc = even ? init : 0.0;
    acc = fma(hm, vm, acc);
    acc = fma(dd, v0, acc);
    acc = fma(hp, vp, acc);
    return even ? acc + 0.0 : init + acc;
}


#define LDS(i) sm[(i)]
#define LSINK(cond, off) seli((cond), (off), L::dmy)
// all rows of the problem, uniform trip count: a lane past the last row recomputes row 0 (rv = false); pure stores rewrite
// row 0 with the same value, read-modify-write stores go to the sink, sums and products are masked with rv
#define LROWS(r, rv, lane_, m_) for (int r0_ = 0; r0_ < (m_); r0_ += WAVE) \
    if (const bool rv = r0_ + (lane_) < (m_); true) if (const int r = rv ? r0_ + (lane_) : 0; true)

// rows c_j(v) for the current iterate; rp = c - t
template <int NMAX, bool DENSE, int MSS>
__device__ __forceinline__ void l_rows(double* sm, const LCtx& x, const crx_lmpc_kparams& kp) {
    using L = LL<NMAX, DENSE, MSS>;
    // straight-line: every lane evaluates all three row kinds on clamped indices and selects; lanes past the last row
    // recompute row 0.  The state rows sum over ALL inputs -- S[k][.][a] is zero for a >= 2k, the bound was a shortcut.
    for (int r0 = 0; r0 < x.m; r0 += WAVE) {
        const int r = r0 + x.lane < x.m ? r0 + x.lane : 0;
        const bool isbox = r < x.r_st, isst = !isbox && r < x.r_lam;
        const int rb = isbox ? r : 0, q = rb & 3;
        const double ub = (q & 2) ? kp.a_max : kp.delta_max, uv = LDS(L::u + 2 * (rb >> 2) + (q >> 1));
        const double cbox = (q & 1) ? ub - uv : uv + ub;
        const int rr = isst ? r - x.r_st : 0, k = rr / 3 + 1, q3 = rr - 3 * (k - 1);
        const int so = L::S + L::srow(k, q3 ? 1 : 0) * L::NU2;
        double s0 = LDS(L::xf + 6 * k + (q3 ? 5 : 0)), s1 = 0.0;
#pragma unroll 4
        for (int a = 0; a < x.nv; a += 2) {
            s0 = fma(LDS(so + a), LDS(L::u + a), s0);
            s1 = fma(LDS(so + a + 1), LDS(L::u + a + 1), s1);
        }
        const double sx = s0 + s1;
        const double cst = q3 == 0 ? kp.v_max - sx : (q3 == 1 ? kp.ey_max - sx : sx + kp.ey_max);
        const double clam = LDS(L::lam + ((!isbox && !isst) ? r - x.r_lam : 0));
        const double cv = sel(isbox, cbox, sel(isst, cst, clam));
        LDS(L::c + r) = cv;
    }
}

// (ru, rl) = g + E'y - J'w   with w = the row array at offset `wo`
template <int NMAX, bool DENSE, int MSS>
__device__ __forceinline__ void l_lagr(double* sm, const LCtx& x, const crx_lmpc_kparams& kp, int wo) {
    using L = LL<NMAX, DENSE, MSS>;
    {
        const bool kv = x.lane >= 1 && x.lane < x.N;
        const int k = kv ? x.lane : 1, r = x.r_st + 3 * (k - 1);
        const double a0 = LDS(wo + r), a1 = LDS(wo + r + 1), a2 = LDS(wo + r + 2);
        LDS(LSINK(kv, L::w0 + k)) = a0;
        LDS(LSINK(kv, L::w5 + k)) = a1 - a2;
    }
    SYNC();
    {
        const bool av = x.lane < x.nv;
        const int a = av ? x.lane : 0, ab = a < x.nu2 ? a : 0, i = ab >> 1, cc = ab & 1;
        double s = LDS(L::gu + a);
#pragma unroll
        for (int c6 = 0; c6 < 6; c6++) s = fma(LDS(L::S + L::srowN(c6) * L::NU2 + a), LDS(L::y + c6), s);
        const double bl = LDS(wo + 4 * i + 2 * cc), bh = LDS(wo + 4 * i + 2 * cc + 1);
        s -= sel(a < x.nu2, bl - bh, 0.0);
        // state rows: c = bound -/+ x_k  ->  J'w = -S0 w_vx - S5 (w_eyhi - w_eylo); all stages (S is zero for k <= a/2)
        double s2 = 0.0;
#pragma unroll 4
        for (int k = 1; k < x.N; k++) {
            s = fma(LDS(L::S + L::srow(k, 0) * L::NU2 + a), LDS(L::w0 + k), s);
            s2 = fma(LDS(L::S + L::srow(k, 1) * L::NU2 + a), LDS(L::w5 + k), s2);
        }
        LDS(LSINK(av, L::ru + a)) = s + s2;
    }
    {
        const bool jv = x.lane < x.M;
        const int j = jv ? x.lane : 0;
        double s = LDS(L::qf + j) + LDS(L::y + 6) - LDS(wo + x.r_lam + j);
#pragma unroll
        for (int c6 = 0; c6 < 6; c6++) s = fma(-LDS(L::SS + c6 * L::MS + j), LDS(L::y + c6), s);
        LDS(LSINK(jv, L::rl + j)) = s;
    }
    SYNC();
}

// NFIX [r3]: the horizon as a compile-time constant (12, the reference's lmpc_param.num_horizon; 0 = read kp.N), as in crx_solve_kernel:
// +3 % on the stand-alone launch and on the closed-loop step (tools/gpu_round3_ac.sh), identical bits (tools/lmpc_ab.py).  The safe-set
// count as a constant too was measured beside it and does not pay.
// [r4] The fixed-horizon instantiation keeps its (compile-time) layout in a STATIC array: the compiler then knows the address (0) and folds
// it into the offset fields of the DS instructions instead of adding a symbol that resolves to 0 at every run-time address (crx_solve_kernel
// has the note): lmpc launch 1.526 -> 1.489 ms, game step 2.76 -> 2.69 ms.  The general instantiations stay on dynamic LDS: static, the
// compiler also sees how many waves the LDS admits per CU, rounds five or six per CU down to ONE per SIMD and takes > 256 registers (four per
// CU); and an amdgpu_waves_per_eu(2) floor against that costs the tuned instantiation 8 % (1.654 ms) -- both measured, tools/gpu_pass.sh game:NAME.
// [r5] The unit is built with MachineLICM off + max-ilp scheduling (Makefile has the numbers): 216 / 222 VGPRs, two waves per SIMD everywhere.
template <int NFIX> struct LmpcStaticLds { static constexpr bool v = CRX_STATIC_LDS && NFIX != 0; };
template <int NMAX, bool DENSE, int MSS, int NFIX = 0>
__global__ void __launch_bounds__(WAVE) crx_lmpc_kernel(const crx_lmpc_kparams kp) {
    static_assert(NFIX <= NMAX, "fixed horizon inside the layout");
    using L = LL<NMAX, DENSE, MSS>;
    double* sm;
    if constexpr (LmpcStaticLds<NFIX>::v) {
        __shared__ __attribute__((aligned(16))) double sm_static[L::END];
        sm = sm_static;
    } else {
        extern __shared__ double sm_dynamic[];
        sm = sm_dynamic;
    }
    if ((int)blockIdx.x >= kp.batch) return;
    const int pb = kp.order ? min(max(kp.order[blockIdx.x], 0), kp.batch - 1) : (int)blockIdx.x;   // dispatch order, see crx_solve_kernel
    if (kp.active && kp.active[pb] == 0) {   // masked launch: this problem is not part of it
        if (threadIdx.x == 0) { kp.status[pb] = CRX_SKIPPED; kp.iters[pb] = 0; }
        return;
    }
    LCtx x;
    x.lane = threadIdx.x;
    x.N = NFIX ? NFIX : kp.N;
    x.nu2 = 2 * x.N;
    x.M = min(max(kp.n_ss[pb], 1), kp.n_ss_max);   // device-resident counts cannot be validated on the host: clamp (M indexes LDS)
    const int lane = x.lane, N = x.N, nu2 = x.nu2, M = x.M, Mx = kp.n_ss_max;
    const crx_ipm_opts& o = kp.opts;
    if (kp.poison) {   // diagnostics (crx_debug_poison_lds)
        for (int e = lane; e < (int)(L::bytes(Mx) / 8); e += WAVE) sm[e] = __longlong_as_double(0x7ff8dead0000beefLL);
        SYNC();
    }

    // ---- load the problem (coalesced) --------------------------------------------------------------
    for (int i = lane; i < 36 * N; i += WAVE) LDS(L::A + i) = kp.A[(size_t)36 * N * pb + i];
    for (int i = lane; i < 12 * N; i += WAVE) LDS(L::B + i) = kp.B[(size_t)12 * N * pb + i];
    for (int i = lane; i < 6 * N; i += WAVE) LDS(L::C + i) = kp.C[(size_t)6 * N * pb + i];
    for (int i = lane; i < 6 * Mx; i += WAVE) {
        int c6 = i / Mx, j = i - c6 * Mx;
        if (j < M) LDS(L::SS + c6 * L::MS + j) = kp.ss[(size_t)6 * Mx * pb + i];
    }
    if (lane < M) LDS(L::qf + lane) = kp.qfun[(size_t)Mx * pb + lane];
    if (lane < 6) LDS(L::xf + lane) = kp.x0[6 * pb + lane];
    const double uold0 = kp.u_old[2 * pb], uold1 = kp.u_old[2 * pb + 1];
    for (int i = lane; i < L::SROWS * L::NU2; i += WAVE) LDS(L::S + i) = 0.0;
    if (DENSE)
        for (int i = lane; i < L::NU2 * L::NU2; i += WAVE) LDS(L::Hu + i) = 0.0;
    SYNC();
    // free response
    for (int k = 0; k < N; k++) {
        if (lane < 6) {
            double s = LDS(L::C + 6 * k + lane);
            for (int c6 = 0; c6 < 6; c6++) s = fma(LDS(L::A + 36 * k + 6 * lane + c6), LDS(L::xf + 6 * k + c6), s);
            LDS(L::xf + 6 * (k + 1) + lane) = s;
        }
        SYNC();
    }
    // cost: f = 1/2 u'Hu u + g0u'u + f0 + qf'lambd; the R / dR part first, the tracking part (Q) stage by stage below
    double f0 = kp.dR[0] * uold0 * uold0 + kp.dR[1] * uold1 * uold1;
    const bool anyQ = kp.Q[0] != 0.0 || kp.Q[1] != 0.0 || kp.Q[2] != 0.0 || kp.Q[3] != 0.0 || kp.Q[4] != 0.0 || kp.Q[5] != 0.0;
    if (lane < nu2) {
        const int a = lane, i = a >> 1, cc = a & 1;
        const double R = cc ? kp.R[1] : kp.R[0], dR = cc ? kp.dR[1] : kp.dR[0];
        double dd = 2.0 * R + 2.0 * dR + (i + 1 < N ? 2.0 * dR : 0.0);
        if (DENSE) {
            LDS(L::Hu + a * L::NU2 + a) = dd;
            if (i > 0) LDS(L::Hu + a * L::NU2 + a - 2) = -2.0 * dR;
            if (i + 1 < N) LDS(L::Hu + a * L::NU2 + a + 2) = -2.0 * dR;
        }
        (void)dd;
        LDS(L::g0u + a) = i == 0 ? -2.0 * dR * (cc ? uold1 : uold0) : 0.0;
    } else if (lane < nu2 + 6) {
        if (DENSE) LDS(L::Hu + lane * L::NU2 + lane) = 2.0 * kp.w_x0;
        LDS(L::g0u + lane) = 0.0;
    }
    if (DENSE && anyQ) {   // (the launcher picks the dense instantiation whenever Q != 0)
        for (int c6 = 0; c6 < 6; c6++) {
            const double r0 = LDS(L::xf + c6) - kp.x_track[c6];
            f0 += kp.Q[c6] * r0 * r0;
        }
        if (lane >= nu2 && lane < nu2 + 6) {   // stage 0 of the second attempt: x_0 = xcurv + w, dx_0/dw = I
            const int c6 = lane - nu2;
            LDS(L::Hu + lane * L::NU2 + lane) += 2.0 * kp.Q[c6];
            LDS(L::g0u + lane) += 2.0 * kp.Q[c6] * (LDS(L::xf + c6) - kp.x_track[c6]);
        }
    }
    SYNC();
    // sensitivities: lane a propagates column a of dx_k/d(u, w) through the stages (columns 2N..2N+5: dx_k/dx_0, used by the
    // second attempt).  Only the rows read later are kept (LL::srow / srowN); the full block of the current stage passes
    // through the scratch TB for the tracking cost.
    {
        const bool col_on = lane < nu2 + 6;
        const int a = col_on ? lane : 0, ka = a >> 1, ca = a & 1;
        double col[6];
        for (int r = 0; r < 6; r++) col[r] = (a >= nu2 && r == a - nu2) ? 1.0 : 0.0;
        for (int k = 0; k < N; k++) {
            double nc[6];
            for (int r = 0; r < 6; r++) {
                double s = (a < nu2 && k == ka) ? LDS(L::B + 12 * k + 2 * r + ca) : 0.0;
                for (int c6 = 0; c6 < 6; c6++) s = fma(LDS(L::A + 36 * k + 6 * r + c6), col[c6], s);
                nc[r] = s;
            }
            for (int r = 0; r < 6; r++) col[r] = nc[r];
            if (col_on) {
                if (k + 1 < N) {
                    LDS(L::S + L::srow(k + 1, 0) * L::NU2 + a) = nc[0];
                    LDS(L::S + L::srow(k + 1, 1) * L::NU2 + a) = nc[5];
                } else {
                    for (int r = 0; r < 6; r++) LDS(L::S + L::srowN(r) * L::NU2 + a) = nc[r];
                }
            }
            if (DENSE && anyQ) {
                if (col_on)
                    for (int r = 0; r < 6; r++) LDS(L::TB + r * L::NU2 + a) = nc[r];
                SYNC();
                for (int c6 = 0; c6 < 6; c6++) {
                    const double q = kp.Q[c6];
                    if (q == 0.0) continue;
                    const double r0 = LDS(L::xf + 6 * (k + 1) + c6) - kp.x_track[c6];
                    if (col_on) {
                        const double sa = nc[c6];
                        LDS(L::g0u + a) += 2.0 * q * r0 * sa;
                        for (int b = 0; b < nu2 + 6; b++) LDS(L::Hu + a * L::NU2 + b) += 2.0 * q * sa * LDS(L::TB + c6 * L::NU2 + b);
                    }
                    f0 += q * r0 * r0;
                }
                SYNC();
            }
        }
    }
    SYNC();
    // rows on the fixed x0 (i = 0) are constants: violated -> the reference's QP is infeasible
    int bad0 = 0;
    {
        const double vx0 = LDS(L::xf + 0), ey0 = LDS(L::xf + 5);
        if (kp.v_max - vx0 < -o.tol || kp.ey_max - ey0 < -o.tol || ey0 + kp.ey_max < -o.tol) bad0 = 1;
    }
    x.r_st = 4 * N;
    x.r_lam = x.r_st + 3 * (N - 1);
    x.r_el = x.r_lam + M;

    // [r3] Reachability screen of the first attempt.  The inputs are boxed, so component c of x_N stays within g_c = sum_a |dx_N,c / du_a|
    // umax_a of its free response (both are on the table after the set-up: xf, the stage-N rows of S), and the terminal equality
    // x_N = SS lambd with lambd in the unit simplex needs x_N,c inside [min_j SS_cj, max_j SS_cj].  Disjoint intervals (by more than
    // 1e-6 of their scale) in ANY component prove that the reference's QP has no feasible point, whatever its other rows say: the
    // first attempt -- 1..10 iterations until the multiplier certificate finds the same -- is skipped and the relaxed second attempt
    // runs as it would have.  Five packed wave reductions per QP; catches 6 of the 11 infeasible recorded QPs; the oracle applies the
    // same test (crx_oracle_lmpc.c).
    // A row violated by the fixed x_0 (bad0: the car is off the track or too fast NOW) makes the reference's QP infeasible before
    // anything is solved: the first attempt used to run to its end and be thrown away -- skipped likewise.
    int first_attempt = (kp.reach_screen && bad0) ? 1 : 0;
    if (kp.reach_screen) {
        double g[6], hi[6], nlo[6];
        const int a = lane < nu2 ? lane : 0, j = lane < M ? lane : 0;
        const double um = (a & 1) ? kp.a_max : kp.delta_max;
#pragma unroll
        for (int c6 = 0; c6 < 6; c6++) {
            g[c6] = sel(lane < nu2, fabs(LDS(L::S + L::srowN(c6) * L::NU2 + a)) * um, 0.0);
            const double sv = LDS(L::SS + c6 * L::MS + j);
            hi[c6] = sel(lane < M, sv, -HUGE_VAL);
            nlo[c6] = sel(lane < M, -sv, -HUGE_VAL);
        }
        double z0 = 0.0, z1 = 0.0;
        wave_sum4(g[0], g[1], g[2], g[3]);
        wave_sum4(g[4], g[5], z0, z1);
        wave_max4(hi[0], hi[1], hi[2], hi[3]);
        wave_max4(hi[4], hi[5], nlo[0], nlo[1]);
        wave_max4(nlo[2], nlo[3], nlo[4], nlo[5]);
        bool out = false;
#pragma unroll
        for (int c6 = 0; c6 < 6; c6++) {
            const double fr = LDS(L::xf + 6 * N + c6), lo = -nlo[c6], tol = 1e-6 * fmax(1.0, fmax(fabs(lo), fabs(hi[c6])));
            out = out || fr - g[c6] > hi[c6] + tol || fr + g[c6] < lo - tol;
        }
        first_attempt |= out ? 1 : 0;   // uniform: every lane holds the same reduced values
    }
    int status = first_attempt ? CRX_INFEASIBLE : CRX_MAX_ITER, total_it = 0;
    int proved = first_attempt | bad0;   // CRX_INFEASIBLE is only ever a PROOF (include/crx.h): the screen / a bound the fixed x_0 violates, or the certificate below
    double E0 = HUGE_VAL, f = 0.0;
    for (int attempt = first_attempt; attempt < 2; attempt++) {
        x.el = attempt;
        x.nv = nu2 + (attempt ? 6 : 0);
        x.m = x.r_el;
        const int m = x.m, nv = x.nv;
        // ---- start point: v = 0, y = 0, t = max(|c|, push), nu = 1 (lambd / elastic rows: cost gradient) ----
        if (lane < L::NU2) LDS(L::u + lane) = 0.0;
        if (lane < L::MS) LDS(L::lam + lane) = 0.0;
        if (lane < 8) LDS(L::y + lane) = 0.0;
        for (int r = lane; r < m; r += WAVE) LDS(L::t + r) = 0.0;
        SYNC();
        l_rows<NMAX, DENSE, MSS>(sm, x, kp);
        SYNC();
        for (int r = lane; r < m; r += WAVE) {
            LDS(L::t + r) = fmax(fabs(LDS(L::c + r)), o.slack_push);
            double nn = 1.0;
            if (r >= x.r_lam) {
                const double gg = LDS(L::qf + (r - x.r_lam));
                if (gg > 1.0) nn = gg;
            }
            LDS(L::nu + r) = nn;
        }
        SYNC();
        double mu = o.mu_init, theta_min = 0.0, theta_max = HUGE_VAL;
        int nf = 0, it = 0;
        status = CRX_MAX_ITER;
        f = f0;
        // sum_j log t_j at the iterate: computed here for the start point, afterwards taken over from the accepted line-search
        // trial (the same slacks) instead of being recomputed every iteration (crx_kernels.hip logsum_t)
        double logsum_t;
        {
            LogAcc la0;
            LROWS(r, rv, lane, m) la0.mul(sel(rv, LDS(L::t + r), 1.0));
            logsum_t = la0.wave_total();
        }
        // stagnation (oracle/crx_oracle_lmpc.c): 25 iterations with the barrier parameter below 1e-6 without reaching tol --
        // the iterate sits on the noise floor of a QP whose unstable local model makes 1e7-sized free responses
        int late = 0;
        for (it = 0;; it++) {
            long long tk[14] = {};
            [[maybe_unused]] int tn = 0;
#ifdef CRX_PHASE_CLOCKS   /* `make TRACE=1`; see crx_kernels.hip */
#define TICK() do { if (kp.trace) tk[tn++] = clock64(); } while (0)
#else
#define TICK() do { } while (0)
#endif
            TICK();
            // ---- rows, gradient, equality residual ----
            l_rows<NMAX, DENSE, MSS>(sm, x, kp);
            {   // gu = g0u + Hu u (lane = row); lanes past nv run row 0 and store to the sink.  Two accumulators.
                const bool av = lane < nv;
                const int a = av ? lane : 0;
                LDS(LSINK(av, L::gu + a)) = hu_dot<L, DENSE>(sm, x, kp, a, L::u, LDS(L::g0u + a));
            }
            {   // e_c = x_N,c - SS_c lambd (c < 6), e_6 = 1'lambd - 1: every lane sums a strided part, one wave sum per row
                double ec[7];
#pragma unroll
                for (int c6 = 0; c6 < 6; c6++) {
                    const int so = L::S + L::srowN(c6) * L::NU2;
                    const int a = lane < nv ? lane : 0, j = lane < M ? lane : 0;
                    const double pu = LDS(so + a) * LDS(L::u + a), pl = LDS(L::SS + c6 * L::MS + j) * LDS(L::lam + j);
                    ec[c6] = sel(lane < nv, pu, 0.0) - sel(lane < M, pl, 0.0);
                }
                ec[6] = sel(lane < M, LDS(L::lam + (lane < M ? lane : 0)), 0.0);
                wave_sum4(ec[0], ec[1], ec[2], ec[3]);     // seven sums in two row reductions (crx_wave.h) instead of seven
                {
                    double zero = 0.0;
                    wave_sum4(ec[4], ec[5], ec[6], zero);
                }
                double ev = ec[6] - 1.0;
#pragma unroll
                for (int c6 = 0; c6 < 6; c6++) ev = sel(lane == c6, ec[c6] + LDS(L::xf + 6 * N + c6), ev);
                LDS(LSINK(lane < 7, L::e + lane)) = ev;
            }
            SYNC();
            TICK();   // 1
            // ---- error measure ----
            l_lagr<NMAX, DENSE, MSS>(sm, x, kp, L::nu);
            double nus = 0.0, e_p = 0.0, e_c = 0.0, theta = 0.0;
            LROWS(r, rv, lane, m) {
                const double tt = LDS(L::t + r), nn = LDS(L::nu + r), rr = fabs(LDS(L::c + r) - tt);
                nus += sel(rv, nn, 0.0);
                e_p = fmax(e_p, rr);
                theta += sel(rv, rr, 0.0);
                e_c = fmax(e_c, tt * nn);
            }
            double ys = 0.0, e_d = 0.0;
            {
                const double ee = fabs(LDS(L::e + (lane < 7 ? lane : 0))), yy = fabs(LDS(L::y + (lane < 7 ? lane : 0)));
                ys = sel(lane < 7, yy, 0.0);
                e_p = fmax(e_p, sel(lane < 7, ee, 0.0));
                theta += sel(lane < 7, ee, 0.0);
                const double ruv = fabs(LDS(L::ru + (lane < nv ? lane : 0))), rlv = fabs(LDS(L::rl + (lane < M ? lane : 0)));
                e_d = fmax(sel(lane < nv, ruv, 0.0), sel(lane < M, rlv, 0.0));
            }
            {
                double zs = 0.0, zm = 0.0;          // (all four maxima are >= 0)
                wave_sum4(nus, ys, theta, zs);
                wave_max4(e_p, e_c, e_d, zm);
            }
            const double sd = fmax(100.0, (nus + ys) / (m + 7)) / 100.0, sc = fmax(100.0, nus / m) / 100.0;
            e_d /= sd;
            e_c /= sc;
            E0 = fmax(e_d, fmax(e_p, e_c));
            // [r6] IPOPT's complete test (crx_kernels.hip has the note): scaled error <= tol AND unscaled dual infeasibility / complementarity within
            // dual_inf_tol / compl_inf_tol (rows are unscaled here: the violation test is implied by e_p <= tol)
            if (E0 <= o.tol && e_d * sd <= o.dual_inf_tol && e_p <= o.constr_viol_tol && e_c * sc <= o.compl_inf_tol) { status = CRX_CONVERGED; break; }
            if (it >= o.max_iter) break;
            if (mu < 1e-6 && ++late >= CRX_LMPC_LATE) break;
            // Still violated: look for the proof that it must be (first attempt only; oracle/crx_oracle_lmpc.c
            // lmpc_certificate()) [r2].  Domain D: inputs in their box, lambd in the unit simplex.  With nu >= 0 on the state
            // rows and ANY y on x_N - SS lambd = 0,  F(v) = sum nu_j c_j(u) - y'e(v)  is linear and >= 0 at every feasible v:
            //     max_D F = F(v) + sum_a (|w_a| ub_a - w_a u_a) + (max_i w_i - sum_i w_i lambd_i) < 0   proves infeasibility.
            // grad F comes out of what l_lagr just stored: w_u = -(ru - gu + nu_lo - nu_hi), w_lambd = qf + y_6 - nu_lambd - rl.
            // A QP that cannot reach the safe set is proven so after 1..10 iterations; the divergence test needed 10..30.
            if (attempt == 0 && it > 0 && theta > 1e-6) {
                double F = 0.0;
                LROWS(r, rv, lane, m) { F += sel(rv && r >= x.r_st && r < x.r_lam, LDS(L::nu + r) * LDS(L::c + r), 0.0); }
                const int q6 = lane < 6 ? lane : 0;
                F -= sel(lane < 6, LDS(L::y + q6) * LDS(L::e + q6), 0.0);
                const bool av = lane < nu2;
                const int a = av ? lane : 0, bi = 4 * (a >> 1) + 2 * (a & 1);
                const double wu = -(LDS(L::ru + a) - LDS(L::gu + a) + LDS(L::nu + bi) - LDS(L::nu + bi + 1));
                F += sel(av, fabs(wu) * ((a & 1) ? kp.a_max : kp.delta_max) - wu * LDS(L::u + a), 0.0);
                const bool jv = lane < M;
                const int j = jv ? lane : 0;
                const double wlam = LDS(L::qf + j) + LDS(L::y + 6) - LDS(L::nu + x.r_lam + j) - LDS(L::rl + j);
                F -= sel(jv, wlam * LDS(L::lam + j), 0.0);
                F = wave_sum(F) + wave_max(sel(jv, wlam, -HUGE_VAL));
                if (F < -1e-8 * (nus + ys)) { status = CRX_INFEASIBLE; break; }
            }
            TICK();   // 2
            // ---- barrier update ----
            for (;;) {
                double e_cm = 0.0;
                LROWS(r, rv, lane, m) { (void)rv; e_cm = fmax(e_cm, fabs(LDS(L::t + r) * LDS(L::nu + r) - mu)); }
                e_cm = wave_max(e_cm) / sc;
                if (fmax(e_d, fmax(e_p, e_cm)) <= o.kappa_eps * mu && mu > o.tol / 10.0) {
                    mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, o.theta_mu == 1.5 ? mu * sqrt(mu) : pow(mu, o.theta_mu)));
                    nf = 0;
                } else
                    break;
            }
            TICK();   // 3
            const double tau = fmax(o.tau_min, 1.0 - mu);
            // ---- Sigma (in dnu), omega = mu/t - Sigma rp (in wv); rhs = -(g + E'y - J'omega) ----
            LROWS(r, rv, lane, m) {
                (void)rv;
                const double tt = LDS(L::t + r), ti = frcp(tt), sg = LDS(L::nu + r) * ti;
                const double rpr = LDS(L::c + r) - tt;
                LDS(L::dnu + r) = sg;
                LDS(L::wv + r) = fma(-sg, rpr, mu * ti);
            }
            SYNC();
            l_lagr<NMAX, DENSE, MSS>(sm, x, kp, L::wv);   // ru, rl = -(rhs)
            // stage weights of the state rows for K_u
            {
                const bool kv = lane >= 1 && lane < N;
                const int k = kv ? lane : 1, r = x.r_st + 3 * (k - 1);
                const double d0 = LDS(L::dnu + r), d1 = LDS(L::dnu + r + 1), d2 = LDS(L::dnu + r + 2);
                LDS(LSINK(kv, L::w0 + k)) = d0;
                LDS(LSINK(kv, L::w5 + k)) = d1 + d2;
            }
            SYNC();
            TICK();   // 4
            // ---- K_u (lower triangle) + extra rows Phi (6) and rhs_u ----
            // One lane per (row a, chunk of 6 columns b0..b0+5): the row's S entries and weights are loaded once per stage
            // and feed six accumulators (an entry-per-lane map re-loaded them for every entry: 3x the LDS reads).  Rows
            // 6g..6g+5 have g+1 chunks each, 3g(g+1) chunks precede group g.  Lanes past the last chunk recompute chunk 0;
            // columns past the diagonal are computed and dropped; the stage loop is uniform (S[k][.][a] = 0 for k <= a/2).
            {
                const int gq = nv / 6, nchunk = 3 * gq * (gq + 1) + (nv - 6 * gq) * (gq + 1);
                for (int c0 = 0; c0 < nchunk; c0 += WAVE) {
                    const int cc = c0 + lane < nchunk ? c0 + lane : 0;
                    const int g = (cc >= 6) + (cc >= 18) + (cc >= 36) + (cc >= 60) + (cc >= 90) + (cc >= 126);
                    const int idx = cc - 3 * g * (g + 1);
                    const int rg = (int)(((float)idx + 0.25f) / (float)(g + 1));
                    const int a = 6 * g + rg, b0 = 6 * (idx - rg * (g + 1));
                    const int ad = a < nu2 ? a : 0;
                    const double d0 = LDS(L::dnu + 4 * (ad >> 1) + 2 * (ad & 1)), d1 = LDS(L::dnu + 4 * (ad >> 1) + 2 * (ad & 1) + 1);
                    double acc[6];
#pragma unroll
                    for (int q = 0; q < 6; q++) acc[q] = hu_entry<L, DENSE>(sm, x, kp, a, b0 + q) + sel(b0 + q == a && a < nu2, d0 + d1, 0.0);
                    for (int k = 1; k < N; k++) {
                        const int r0 = L::S + L::srow(k, 0) * L::NU2, r5 = L::S + L::srow(k, 1) * L::NU2;
                        const double sa0 = LDS(r0 + a), sa5 = LDS(r5 + a), w0k = LDS(L::w0 + k), w5k = LDS(L::w5 + k);
                        double sb0[6], sb5[6];
#pragma unroll
                        for (int q = 0; q < 6; q++) { sb0[q] = LDS(r0 + b0 + q); sb5[q] = LDS(r5 + b0 + q); }
                        __builtin_amdgcn_sched_barrier(0);
                        const double t0 = sa0 * w0k, t5 = sa5 * w5k;
#pragma unroll
                        for (int q = 0; q < 6; q++) acc[q] = fma(t5, sb5[q], fma(t0, sb0[q], acc[q]));
                    }
#pragma unroll
                    for (int q = 0; q < 6; q++) LDS(LSINK(b0 + q <= a, L::K + a * L::LDK + b0 + q)) = acc[q];
                }
            }
            for (int e0 = 0; e0 < 7 * nv; e0 += WAVE) {
                const int en = e0 + lane < 7 * nv ? e0 + lane : 0;
                const int r = en / nv, j = en - r * nv;
                const double sv = LDS(L::S + L::srowN(r < 6 ? r : 0) * L::NU2 + j), rv = LDS(L::ru + j);
                LDS(L::K + (nv + r) * L::LDK + j) = sel(r < 6, sv, -rv);
            }
            SYNC();
            TICK();   // 5
            int ok = l_chol(sm, L::K, L::LDK, L::ik, nv, 7, lane, L::IKS);
            TICK();   // 6
            if (!ok) break;
            // ---- W~ and b_x ----
            if (lane < 27) {
                double s = 0.0;
                if (lane < 21) {
                    int r = 0, q = lane;
                    while (q > r) { q -= r + 1; r++; }   // lane -> (r, q), q <= r
                    const int ro = L::K + (nv + r) * L::LDK, qo = L::K + (nv + q) * L::LDK;
#pragma unroll 8
                    for (int j = 0; j < nv; j++) s = fma(LDS(ro + j), LDS(qo + j), s);
                    LDS(L::Wt + 6 * r + q) = s;
                    LDS(L::Wt + 6 * q + r) = s;
                } else {
                    const int r = lane - 21;
                    s = LDS(L::e + r);
                    const int ro = L::K + (nv + r) * L::LDK, zo = L::K + (nv + 6) * L::LDK;
#pragma unroll 8
                    for (int j = 0; j < nv; j++) s = fma(LDS(ro + j), LDS(zo + j), s);
                    LDS(L::bx + r) = s;
                }
            }
            SYNC();
            // ---- L_w (6x6, per lane), T_j = L_w^-1 ss_j, tbx = L_w^-1 b_x ----
            double Lw[6][6], iw[6], tbx[6], Tj[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
#pragma unroll
                for (int i = j; i < 6; i++) {
                    double s = LDS(L::Wt + 6 * i + j);
#pragma unroll
                    for (int k = 0; k < j; k++) s = fma(-Lw[i][k], Lw[j][k], s);
                    Lw[i][j] = s;
                }
                if (!(Lw[j][j] > 0.0)) ok = 0;
                iw[j] = frsqrt(fmax(Lw[j][j], 1e-300));
#pragma unroll
                for (int i = j; i < 6; i++) Lw[i][j] *= iw[j];
            }
            if (!ok) break;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double s = LDS(L::bx + i), sj = sel(lane < M, LDS(L::SS + i * L::MS + (lane < M ? lane : 0)), 0.0);
#pragma unroll
                for (int k = 0; k < i; k++) {
                    s = fma(-Lw[i][k], tbx[k], s);
                    sj = fma(-Lw[i][k], Tj[k], sj);
                }
                tbx[i] = s * iw[i];
                Tj[i] = sj * iw[i];
            }
            // ---- G = D_lambda + T T' is diagonal plus rank six: factorised in product form, never assembled.
            //   G = L_1 ... L_6 D' L_6' ... L_1',  L_c = I + strict_lower(p_c beta_c'),  one rank-one update of the diagonal
            //   factor per column tau_c of T (Gill-Golub-Murray-Saunders C1, positive updates):
            //     p = (L_1..L_{c-1})^-1 tau_c,  t_j = 1 + sum_{i<=j} p_i^2/d_i,  d'_j = d_j t_j/t_{j-1},  beta_j = p_j/(d_j t_j).
            //   With pd = p/d and pt_j = p_j/t_{j-1} both triangular solves collapse to ONE scan each:
            //     L_c^-1 b = b - pt .* exclusive_prefix(pd .* b),   L_c^-T b = b - pd .* exclusive_suffix(pt .* b)
            //   (lane j = element j, M <= 60): 21 scans to factorise, 12 scan steps for the two right-hand sides, instead
            //   of a 44 x 44 Cholesky (30 % of the iteration), its assembly and its back substitution.  All sums are of
            //   one sign or short; measured componentwise backward error 4e-14 for d spanning 1e-16..1e16 (dense
            //   Cholesky: 8e-15), tools/ubench note in DESIGN.md section 5.3.  The oracle keeps the dense Cholesky.
            double b2[2];
            {
                const bool lv = lane < M;
                const int lc = lv ? lane : 0;
                double s = -LDS(L::rl + lc);
#pragma unroll
                for (int c6 = 0; c6 < 6; c6++) s = fma(Tj[c6], tbx[c6], s);
                double dcur = sel(lv, LDS(L::dnu + x.r_lam + lc), 1.0);
                if (__any(!(dcur > 0.0))) ok = 0;
                double pd[6], pt[6], pcol[6];
#pragma unroll
                for (int c6 = 0; c6 < 6; c6++) pcol[c6] = sel(lv, Tj[c6], 0.0);
#pragma unroll
                for (int c6 = 0; c6 < 6; c6++) {
                    const double pc = pcol[c6];
                    const double pdv = pc * frcp(dcur), w = pc * pdv;
                    const double tprev = 1.0 + excl_prefix(w, lane), rtp = frcp(tprev);
                    pd[c6] = pdv;
                    pt[c6] = pc * rtp;
                    dcur = dcur * (tprev + w) * rtp;
                    // L_c^-1 on the columns still to come: independent scans, in flight together (depth 2 per column of T
                    // instead of c + 1)
#pragma unroll
                    for (int cc = c6 + 1; cc < 6; cc++) pcol[cc] = fma(-pt[c6], excl_prefix(pdv * pcol[cc], lane), pcol[cc]);
                }
                const double rdf = frcp(dcur);
                double y0 = sel(lv, s, 0.0), y1 = sel(lv, 1.0, 0.0);
#pragma unroll
                for (int c6 = 0; c6 < 6; c6++) {
                    const double e0 = excl_prefix(pd[c6] * y0, lane), e1 = excl_prefix(pd[c6] * y1, lane);
                    y0 = fma(-pt[c6], e0, y0);
                    y1 = fma(-pt[c6], e1, y1);
                }
                y0 *= rdf;
                y1 *= rdf;
#pragma unroll
                for (int c6 = 5; c6 >= 0; c6--) {
                    const double e0 = excl_suffix(pt[c6] * y0, lane), e1 = excl_suffix(pt[c6] * y1, lane);
                    y0 = fma(-pd[c6], e0, y0);
                    y1 = fma(-pd[c6], e1, y1);
                }
                b2[0] = y0;
                b2[1] = y1;
            }
            TICK();   // 7
            TICK();   // 8
            TICK();   // 9
            if (!ok) break;
            {
                double s1 = sel(lane < M, b2[0], 0.0), s2 = sel(lane < M, b2[1], 0.0);
                wave_sum2(s1, s2);
                const double dy1 = (s1 + LDS(L::e + 6)) / s2;
                const double dl = sel(lane < M, b2[0] - b2[1] * dy1, 0.0);
                LDS(LSINK(lane < M, L::dlam + lane)) = dl;
                double tb[6], dyx[6];
                {
                    double w6[6];
#pragma unroll
                    for (int c6 = 0; c6 < 6; c6++) w6[c6] = Tj[c6] * dl;
                    wave_sum4(w6[0], w6[1], w6[2], w6[3]);
                    wave_sum2(w6[4], w6[5]);
#pragma unroll
                    for (int c6 = 0; c6 < 6; c6++) tb[c6] = tbx[c6] - w6[c6];
                }
#pragma unroll
                for (int i = 5; i >= 0; i--) {
                    double s = tb[i];
#pragma unroll
                    for (int k = i + 1; k < 6; k++) s = fma(-Lw[k][i], dyx[k], s);
                    dyx[i] = s * iw[i];
                }
                if (lane == 0) {
#pragma unroll
                    for (int c6 = 0; c6 < 6; c6++) LDS(L::dy + c6) = dyx[c6];
                    LDS(L::dy + 6) = dy1;
                }
                // du = L_u^-T (z - Y' dy_x)
                double b1[1];
                {
                    const int a = lane < nv ? lane : 0;
                    double s = LDS(L::K + (nv + 6) * L::LDK + a);
#pragma unroll
                    for (int r = 0; r < 6; r++) s = fma(-LDS(L::K + (nv + r) * L::LDK + a), dyx[r], s);
                    b1[0] = sel(lane < nv, s, 0.0);
                }
                l_backsub<1>(sm, L::K, L::LDK, L::ik, nv, lane, b1, L::IKS);
                LDS(LSINK(lane < nv, L::du + lane)) = b1[0];
            }
            SYNC();
            TICK();   // 10
            // ---- row steps ----
            double rp_max = 0.0, rd_max = 0.0, Dphi = 0.0;
            LROWS(r, rv, lane, m) {
                const bool isbox = r < x.r_st, isst = !isbox && r < x.r_lam;
                const int rb = isbox ? r : 0;
                const double db = LDS(L::du + 2 * (rb >> 2) + ((rb & 3) >> 1));
                const double jbox = (rb & 1) ? -db : db;
                const int rr = isst ? r - x.r_st : 0, k = rr / 3 + 1, q = rr - 3 * (k - 1), so = L::S + L::srow(k, q ? 1 : 0) * L::NU2;
                double s0 = 0.0, s1 = 0.0;     // all inputs: S is zero for a >= 2k
#pragma unroll 4
                for (int a = 0; a < nv; a += 2) {
                    s0 = fma(LDS(so + a), LDS(L::du + a), s0);
                    s1 = fma(LDS(so + a + 1), LDS(L::du + a + 1), s1);
                }
                const double sx = s0 + s1, jst = q == 2 ? sx : -sx;
                const double jlam = LDS(L::dlam + ((!isbox && !isst) ? r - x.r_lam : 0));
                const double jd = sel(isbox, jbox, sel(isst, jst, jlam));
                const double tt = LDS(L::t + r), nn = LDS(L::nu + r), ti = frcp(tt);
                const double dtt = (LDS(L::c + r) - tt) + jd;
                const double dn = (mu - tt * nn - nn * dtt) * ti;
                LDS(L::wv + r) = jd;
                rp_max = fmax(rp_max, -dtt * ti);
                rd_max = fmax(rd_max, -dn * frcp(nn));
                Dphi = fma(sel(rv, -mu * dtt, 0.0), ti, Dphi);
                LDS(L::dnu + r) = dn;   // the multiplier step, in the array Sigma has left (dead after the K_u assembly)
            }
            double gdv = 0.0, qd = 0.0;
            {
                const bool av = lane < nv;
                const int a = av ? lane : 0, j = lane < M ? lane : 0;
                const double d = LDS(L::du + a), gua = LDS(L::gu + a);
                qd = sel(av, hu_dot<L, DENSE>(sm, x, kp, a, L::du, 0.0) * d, 0.0);
                gdv = sel(av, gua * d, 0.0) + sel(lane < M, LDS(L::qf + j) * LDS(L::dlam + j), 0.0);
            }
            wave_max2(rp_max, rd_max);
            double esum = sel(lane < 7, fabs(LDS(L::e + (lane < 7 ? lane : 0))), 0.0);
            wave_sum4(gdv, qd, Dphi, esum);
            Dphi += gdv;
            const double a_p = rp_max > tau ? tau / rp_max : 1.0, a_d = rd_max > tau ? tau / rd_max : 1.0;
            const double phi0 = f - mu * logsum_t;
            if (it == 0) {
                theta_min = 1e-4 * fmax(1.0, theta);
                theta_max = 1e4 * fmax(1.0, theta);
            }
            TICK();   // 11
            // ---- filter line search (all rows linear: c(v + al dv) = c + al J dv) ----
            double al = a_p, fn = f, lt = logsum_t;
            int acc = 0, ftype = 0;
            for (int ls = 0; ls < 40 && al >= 1e-10; ls++) {   // alpha_min: see crx_kernels.hip
                fn = f + al * (gdv + 0.5 * al * qd);
                double thn = 0.0;
                LogAcc la;
                LROWS(r, rv, lane, m) {
                    const double tt = LDS(L::t + r), jd = LDS(L::wv + r), rpr = LDS(L::c + r) - tt;
                    const double cj = (rpr + tt) + al * jd;
                    double tn = fma(al, rpr + jd, tt);
                    tn = fmax(tn, cj);
                    la.mul(sel(rv, tn, 1.0));
                    thn += sel(rv, fabs(cj - tn), 0.0);
                }
                lt = la.wave_total_with(thn);                // thn and the exponent sum share one reduction
                thn += (1.0 - al) * esum;
                const double phin = fn - mu * lt;
                int okf = (thn <= theta_max) && (phin == phin);
                for (int i = 0; i < nf && okf; i++)
                    if (!(thn < LDS(L::Fth + i) || phin < LDS(L::Fph + i))) okf = 0;
                if (okf) {
                    // switching test al (-Dphi)^2.3 > theta^1.1 in the log2 domain (crx_kernels.hip: two pow() are ~2.6 k cycles)
                    const int sw = (Dphi < 0.0) && (log2_fast(al) + 2.3 * log2_fast(-Dphi) > 1.1 * log2_fast(theta));
                    if (theta <= theta_min && sw) {
                        if (phin <= phi0 + 1e-8 * al * Dphi + 10.0 * 2.2e-16 * fabs(phi0)) { acc = 1; ftype = 1; }
                    } else if (thn <= (1.0 - 1e-5) * theta || phin <= phi0 - 1e-8 * theta) {
                        acc = 1;
                    }
                }
                if (acc) break;
                al *= 0.5;
            }
            if (acc && !ftype && nf < LMAXF) {
                if (lane == 0) {
                    LDS(L::Fth + nf) = (1.0 - 1e-5) * theta;
                    LDS(L::Fph + nf) = phi0 - 1e-8 * theta;
                }
                nf++;
            }
            if (!acc) break;
            TICK();   // 12
            // ---- accept ----
            logsum_t = lt;
            double numax = 0.0;
            LROWS(r, rv, lane, m) {
                const double tt = LDS(L::t + r), jd = LDS(L::wv + r), rpr = LDS(L::c + r) - tt;
                const double cj = (rpr + tt) + al * jd;
                const double tn = fmax(fma(al, rpr + jd, tt), cj);
                double nn = fma(a_d, LDS(L::dnu + r), LDS(L::nu + r));
                const double mut = mu * frcp(tn);
                nn = fmin(fmax(nn, mut * 1e-10), mut * 1e10);
                LDS(LSINK(rv, L::t + r)) = tn;          // read-modify-write
                LDS(LSINK(rv, L::nu + r)) = nn;
                numax = fmax(numax, sel(rv, nn, 0.0));
            }
            {
                const int a = lane < nv ? lane : 0, j = lane < M ? lane : 0, q = lane < 7 ? lane : 0;
                const double un = fma(al, LDS(L::du + a), LDS(L::u + a)), ln = fma(al, LDS(L::dlam + j), LDS(L::lam + j));
                const double yn = fma(al, LDS(L::dy + q), LDS(L::y + q));
                LDS(LSINK(lane < nv, L::u + a)) = un;
                LDS(LSINK(lane < M, L::lam + j)) = ln;
                LDS(LSINK(lane < 7, L::y + q)) = yn;
            }
            f = fn;
            numax = wave_max(numax);
            SYNC();
            TICK();   // 13
            if (kp.trace && pb == kp.trace_problem && lane == 0 && total_it + it < kp.trace_rows) {
                double* tr = kp.trace + (size_t)(total_it + it) * 16;
                for (int q = 0; q < 13; q++) tr[q] = (double)(tk[q + 1] - tk[q]);
                tr[13] = (double)(tk[13] - tk[0]);
                tr[14] = mu;
                tr[15] = al;
            }
            if (numax > 1e12 && theta > 1e-6) { status = CRX_STALLED; it++; break; }   // IPOPT's divergence heuristic: not a proof
        }
        total_it += it;
        SYNC();
        if (attempt == 0 && status == CRX_CONVERGED && !bad0) break;
        if (attempt == 0 && status == CRX_INFEASIBLE) proved = 1;
        // the reference's (pinned) QP was not solved; without a proof of its infeasibility (divergence heuristic, iteration cap, stagnation) the
        // relaxed plan is reported CRX_STALLED: same plan, no claim about the pinned QP
        if (attempt == 1 && status == CRX_CONVERGED) status = proved ? CRX_INFEASIBLE : CRX_STALLED;
    }
    // ---- write back: the plan by a roll-out of the model from x_0 (+ w in the second attempt) with the optimal inputs;
    // A, B, C come back from HBM into the K region, the states go through the scratch TB ----
    for (int i = lane; i < 36 * N; i += WAVE) LDS(L::A + i) = kp.A[(size_t)36 * N * pb + i];
    for (int i = lane; i < 12 * N; i += WAVE) LDS(L::B + i) = kp.B[(size_t)12 * N * pb + i];
    for (int i = lane; i < 6 * N; i += WAVE) LDS(L::C + i) = kp.C[(size_t)6 * N * pb + i];
    if (lane < 6) LDS(L::TB + lane) = LDS(L::xf + lane) + (x.nv > nu2 ? LDS(L::u + nu2 + lane) : 0.0);
    SYNC();
    for (int k = 0; k < N; k++) {
        if (lane < 6) {
            double s = LDS(L::C + 6 * k + lane);
            for (int c6 = 0; c6 < 6; c6++) s = fma(LDS(L::A + 36 * k + 6 * lane + c6), LDS(L::TB + 6 * k + c6), s);
            s = fma(LDS(L::B + 12 * k + 2 * lane), LDS(L::u + 2 * k), s);
            s = fma(LDS(L::B + 12 * k + 2 * lane + 1), LDS(L::u + 2 * k + 1), s);
            LDS(L::TB + 6 * (k + 1) + lane) = s;
        }
        SYNC();
    }
    for (int i = lane; i < 6 * (N + 1); i += WAVE) kp.X[(size_t)6 * (N + 1) * pb + i] = LDS(L::TB + i);
    if (lane < nu2) kp.U[(size_t)nu2 * pb + lane] = LDS(L::u + lane);
    if (lane < Mx) kp.lambda[(size_t)Mx * pb + lane] = lane < M ? LDS(L::lam + lane) : 0.0;
    if (lane == 0) {
        double ql = 0.0;
        for (int j = 0; j < M; j++) ql = fma(LDS(L::qf + j), LDS(L::lam + j), ql);
        (void)ql;
        kp.cost[pb] = f;
        kp.status[pb] = status;
        kp.kkt[pb] = E0;
        kp.iters[pb] = total_it;
    }
}

template <int NMAX, bool DENSE, int MSS = CRX_MAX_SS, int NFIX = 0>
static hipError_t launch_l(const crx_lmpc_kparams& kp, hipStream_t st) {
    if constexpr (LmpcStaticLds<NFIX>::v) {
        hipLaunchKernelGGL((crx_lmpc_kernel<NMAX, DENSE, MSS, NFIX>), dim3(kp.batch), dim3(WAVE), 0, st, kp);   // the layout is a static array of the kernel
        return hipGetLastError();
    }
    const size_t bytes = LL<NMAX, DENSE, MSS>::bytes(kp.n_ss_max);
    // the opt-in to > 64 KiB of dynamic LDS is a property of the (function, device) pair: set once per device
    static int attr_set_on = -1;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (attr_set_on != dev) {
        hipError_t e = hipFuncSetAttribute((const void*)crx_lmpc_kernel<NMAX, DENSE, MSS, NFIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        attr_set_on = dev;
    }
    hipLaunchKernelGGL((crx_lmpc_kernel<NMAX, DENSE, MSS, NFIX>), dim3(kp.batch), dim3(WAVE), bytes, st, kp);
    return hipGetLastError();
}

// Q == 0 (no tracking cost: the reference's LMPCRacingParam default): the instantiation that does not store Hu
static bool lmpc_dense(const crx_lmpc_kparams& kp) {
    for (int c = 0; c < 6; c++)
        if (kp.Q[c] != 0.0) return true;
    return false;
}

hipError_t crx_launch_lmpc(const crx_lmpc_kparams& kp, hipStream_t st) {
    if (kp.batch == 0) return hipSuccess;
    if (lmpc_dense(kp)) return kp.N <= 12 ? launch_l<12, true>(kp, st) : launch_l<CRX_LMPC_MAX_N, true>(kp, st);
    // the reference's configuration (N = 12, Q = 0, 44 safe-set points: utils/base.py:350-376) has its own instantiation: six per CU
    if (kp.N == 12 && kp.n_ss_max <= CRX_LMPC_SS44) return launch_l<12, false, CRX_LMPC_SS44, 12>(kp, st);
    if (kp.N <= 12 && kp.n_ss_max <= CRX_LMPC_SS44) return launch_l<12, false, CRX_LMPC_SS44>(kp, st);
    return kp.N <= 12 ? launch_l<12, false>(kp, st) : launch_l<CRX_LMPC_MAX_N, false>(kp, st);
}

template <int NMAX, bool DENSE, int MSS = CRX_MAX_SS, int NFIX = 0>
static int occ_l(int n_ss_max) {
    int n = 0;
    const size_t bytes = LmpcStaticLds<NFIX>::v ? 0 : LL<NMAX, DENSE, MSS>::bytes(n_ss_max);   // static LDS: the runtime counts it by itself
    if (!LmpcStaticLds<NFIX>::v &&
        hipFuncSetAttribute((const void*)crx_lmpc_kernel<NMAX, DENSE, MSS, NFIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, crx_lmpc_kernel<NMAX, DENSE, MSS, NFIX>, WAVE, bytes) != hipSuccess) return -1;
    return n;
}
// (diagnostics: the Q == 0 instantiation the launcher would pick for (N, n_ss_max) -- the reference's parameters select it; the
// same selection as crx_launch_lmpc, fixed-horizon instantiation included: the occupancy of the kernel that is timed)
int crx_lmpc_resident_per_cu(int N, int n_ss_max) {
    if (N == 12 && n_ss_max <= CRX_LMPC_SS44) return occ_l<12, false, CRX_LMPC_SS44, 12>(n_ss_max);
    if (N <= 12 && n_ss_max <= CRX_LMPC_SS44) return occ_l<12, false, CRX_LMPC_SS44>(n_ss_max);
    return N <= 12 ? occ_l<12, false>(n_ss_max) : occ_l<CRX_LMPC_MAX_N, false>(n_ss_max);
}

size_t crx_lmpc_lds_bytes(int N, int n_ss_max) {
    if (N <= 12 && n_ss_max <= CRX_LMPC_SS44) return LL<12, false, CRX_LMPC_SS44>::bytes(n_ss_max);
    return N <= 12 ? LL<12, false>::bytes(n_ss_max) : LL<CRX_LMPC_MAX_N, false>::bytes(n_ss_max);
}
static_assert(LL<12, false, CRX_LMPC_SS44>::bytes(CRX_LMPC_SS44) <= 27264, "six learning-MPC QPs per CU");
static_assert(LL<CRX_LMPC_MAX_N, true>::bytes(CRX_MAX_SS) <= 160 * 1024, "LDS budget");
