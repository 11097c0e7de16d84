// crx_lmpcprep.hip -- the host work in front of the learning-MPC QP, on the device:
//   crx_lmpc_prep_kernel      per race: N local LTV stage models (kernel-weighted least squares on the two previous laps +
//                             analytic Jacobian of the Euler step) and the safe-set points / cost-to-go of the QP
//   crx_lmpc_addpoint_kernel  LMPCRacingGame.add_point: the running lap extends the previous lap's safe set
//
// Restates (paths into /root/reference/car_racing):
//   control/lmpc_helper.py:26-189   regression_and_linearization     :192-226  compute_index
//                      :229-275     compute_Q_M                      :338-355  compute_b
//                      :358-366     lmpc_loc_lin_reg (cvxopt qp without constraints = a 5x5 linear solve)
//                      :278-293     select_points
//   utils/base.py:585-622 estimate_ABC, :624-629 add_point;  control/control.py:625-639, :726-728
//
// One wavefront per race.  The five regression features (vx, vy, wz, delta, a) of the two laps used are staged in LDS
// once (lanes read consecutive samples of their race's safe set: coalesced), then per stage: distances (lane per sample),
// rank selection of the max_neighbours nearest (lane per sample, broadcast reads of the distances), compaction by ballot,
// and the 45 sums of the two normal matrices and three right-hand sides, ONE LANE PER SUM, each running over the selected
// samples in ascending index order -- the order, and the absence of fused multiply-adds, are those of
// oracle/crx_oracle_lmpc_prep.c: with normal matrices of condition 3e11 that is what makes kernel and oracle agree to
// the last bits instead of to 1e-5.  HBM-bound in principle (reads ~2 n (6+2) doubles of safe set per race once, writes
// N 54 + 7 M doubles), a few percent of the QP solve it feeds.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "crx_kparams.h"
#include "crx_wave.h"

#pragma clang fp contract(off)

// the reference's get_curvature: s wrapped into one lap, first segment with lo <= s <= hi
__device__ static double lp_curvature(const double* track, int n_seg, double lap_length, double s) {
    s = wrap_below(wrap_above(s, lap_length), lap_length);
    for (int i = 0; i < n_seg; i++)
        if (s >= track[6 * i + 3] && s <= track[6 * i + 3] + track[6 * i + 4]) return track[6 * i + 5];
    return 0.0;
}

// Gaussian elimination with partial pivoting, 5x5, nrhs right-hand sides; every lane runs it on the same numbers.
// Operation for operation oracle/crx_oracle_lmpc_prep.c solve5().
template <int NRHS>
__device__ __forceinline__ bool lp_solve5(double (&Q)[5][5], double (&rhs)[NRHS][5]) {
    // every index below is a compile-time constant after unrolling (the pivot row is applied through selects over the
    // candidate rows): the matrices stay in registers.  With a run-time row index they lived in scratch memory and the
    // 24 solves of a race cost ~1 ms of dependent scratch round trips -- 45 % of a whole control step of the batched lap.
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        int p = c;
        double best = fabs(Q[c][c]);
#pragma unroll
        for (int r = c + 1; r < 5; r++) {
            const double a = fabs(Q[r][c]);
            const bool gt = a > best;
            best = gt ? a : best;
            p = gt ? r : p;
        }
        if (!(best > 0.0) || !isfinite(best)) ok = false;
#pragma unroll
        for (int r = c + 1; r < 5; r++) {
            const bool sw = p == r;
#pragma unroll
            for (int k = 0; k < 5; k++) { const double t = Q[c][k], u = Q[r][k]; Q[c][k] = sw ? u : t; Q[r][k] = sw ? t : u; }
#pragma unroll
            for (int q = 0; q < NRHS; q++) { const double t = rhs[q][c], u = rhs[q][r]; rhs[q][c] = sw ? u : t; rhs[q][r] = sw ? t : u; }
        }
#pragma unroll
        for (int r = c + 1; r < 5; r++) {
            const double f = Q[r][c] / Q[c][c];
#pragma unroll
            for (int k = c; k < 5; k++) { const double t = f * Q[c][k]; Q[r][k] = Q[r][k] - t; }
#pragma unroll
            for (int q = 0; q < NRHS; q++) { const double t = f * rhs[q][c]; rhs[q][r] = rhs[q][r] - t; }
        }
    }
#pragma unroll
    for (int q = 0; q < NRHS; q++)
#pragma unroll
        for (int r = 4; r >= 0; r--) {
            double s = rhs[q][r];
#pragma unroll
            for (int k = r + 1; k < 5; k++) { const double t = Q[r][k] * rhs[q][k]; s = s - t; }
            rhs[q][r] = s / Q[r][r];
        }
    return ok;
}

// LDS: feat [2][P][5] | per wave: dist [P] | wsel [2*MAXNB] | sums [48] | isel int[2*MAXNB]
#define LP_MAXNB 64
// LP_WAVES wavefronts per race [r2]: the stages are independent (each has its own linearisation point), wave w takes
// stages w, w + LP_WAVES, ...; the feature table is staged once by all of them and shared, every wave has its own
// distance / selection / sums scratch, so that past the staging barrier the waves never meet again (SYNC() stays a
// wavefront fence) until the status word is collected.  One wave per race left a CU with two or three waves in flight,
// each running 12 stages of dependent LDS round trips one after the other: 1.15 ms per 1024 races; with four waves
// per race (three stages each, eight waves per CU) the same arithmetic, bit for bit, takes a third of that.
// [r3] More waves per race were measured again (tools/gpu_round3_o.sh, 4096 races, the kernel alone): 6 -> 1.15 ms, 8 -> 1.08,
// 12 (one stage per wave, one 129 KB workgroup per CU) -> 0.75, 16 -> 0.88 against 0.875 ms here; the twelve-wave build moves a
// closed-loop step by 1 % (3.32 -> 3.27 ms learning-MPC laps, racing game unchanged: the big workgroup overlaps less with the
// other stream's kernels) and idles waves for N < 12 -- left at four.
#ifndef LP_WAVES
#define LP_WAVES 4
#endif
#define LP_WAVE_DOUBLES(P_) ((size_t)(P_) + 2 * LP_MAXNB + 48 + LP_MAXNB /* isel: 2*MAXNB ints */)

__global__ void __launch_bounds__(WAVE * LP_WAVES) crx_lmpc_prep_kernel(const crx_lmpcprep_kparams kp) {
    extern __shared__ __attribute__((aligned(16))) double lsm[];
    __shared__ int sbad;
    const crx_lmpcprep_desc& d = kp.d;
    const int b = blockIdx.x, lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    if (b >= kp.batch) return;
    if (kp.active && kp.active[b] == 0) {   // masked launch: this race is not part of it (uniform over the workgroup)
        if (threadIdx.x == 0) kp.status[b] = CRX_SKIPPED;
        return;
    }
    const int N = d.N, P = d.n_points, L = d.n_laps, M = d.n_ss_per_lap * d.n_ss_laps;
    double* feat = lsm;
    double* dist = feat + (size_t)2 * P * 5 + (size_t)wv * LP_WAVE_DOUBLES(P);
    double* wsel = dist + P;
    double* sums = wsel + 2 * LP_MAXNB;
    int* isel = (int*)(sums + 48);
    if (threadIdx.x == 0) sbad = 0;
    const double* ss = kp.ss_xcurv + (size_t)b * L * P * 6;
    const double* us = kp.u_ss + (size_t)b * L * P * 2;
    const double* qf = kp.qfun + (size_t)b * L * P;
    int it = kp.iter[b];
    it = it < 2 ? 2 : (it > L ? L : it);   // device-resident counts cannot be validated on the host: clamp
    int nl[2];
    for (int lapk = 0; lapk < 2; lapk++) {
        const int lap = it - 2 + lapk;
        int n = kp.time_ss[(size_t)b * L + lap] - 1;
        n = n < 1 ? 1 : (n > P - 1 ? P - 1 : n);
        nl[lapk] = n;
        for (int e = threadIdx.x; e < (n + 1) * 5; e += WAVE * LP_WAVES) {   // rows 0..n (row j+1 is the regression target of row j)
            const int j = e / 5, c = e - 5 * j;
            feat[((size_t)lapk * P + j) * 5 + c] = c < 3 ? ss[((size_t)lap * P + j) * 6 + c] : us[((size_t)lap * P + j) * 2 + (c - 3)];
        }
    }
    __syncthreads();
    int bad = 0;
    for (int i = wv; i < N; i += LP_WAVES) {
        // linearisation point: given, or the previous plan shifted by one stage (control.py:726-728)
        const int ix = kp.from_plan ? (i + 1 <= N ? i + 1 : N) : i, iu = kp.from_plan ? (i + 1 <= N - 1 ? i + 1 : N - 1) : i;
        double x0[6], u0[2];
        for (int k = 0; k < 6; k++) x0[k] = kp.lin_points[((size_t)b * (N + 1) + ix) * 6 + k];
        for (int k = 0; k < 2; k++) u0[k] = kp.lin_input[((size_t)b * N + iu) * 2 + k];
        const double xl[5] = {x0[0], x0[1], x0[2], u0[0], u0[1]};
        int nsel_tot = 0;
        for (int lapk = 0; lapk < 2; lapk++) {
            const int n = nl[lapk];
            const double* F = feat + (size_t)lapk * P * 5;
            int inside = 0;
            for (int j0 = 0; j0 < n; j0 += WAVE) {
                const int j = j0 + lane;
                double s = INFINITY;
                if (j < n) {
                    s = 0.0;
                    for (int c = 0; c < 5; c++) s = s + fabs((F[j * 5 + c] - xl[c]) * d.scale[c]);
                    dist[j] = s;
                }
                inside += __popcll(__ballot(s < d.bandwidth));
            }
            SYNC();
            // selected set: everything inside the bandwidth, or the max_neighbours nearest by (distance, index) -- the oracle's
            // `rank < max_neighbours`.  Ranking every sample against every other is 2 M double compares per race and step
            // (4 ms per step of 1024 races); instead the max_neighbours-th smallest distance is found by bisection on the
            // bit pattern of the (non-negative) distances -- at most 63 counting passes -- and ties at that value are taken in
            // ascending index order.  Compaction keeps ascending index order.  [r2] The search starts below the bandwidth (the
            // k-th smallest is inside it) and STOPS as soon as a threshold has exactly k keys at or below it: those are the k
            // nearest, which threshold between the k-th and the (k+1)-th key it was does not matter (v* = threshold + 1 selects
            // `key < v*`, no tie is taken) -- ~20 passes instead of 63; only a tie AT the k-th distance runs the search to the
            // end.  The bisection was 54 % of the kernel (tools: build with the selection disabled).
            int nsel = 0;
            const bool top = inside >= d.max_neighbours;
            unsigned long long vstar = 0x7ff0000000000000ull;            // +inf: nothing is cut
            int n_less = 0;
            if (top) {
                unsigned long long lo = 0ull, hi = (unsigned long long)__double_as_longlong(d.bandwidth);   // inside >= k: the k-th smallest is < bandwidth
                constexpr int KP = 8;                                     // laps of up to 512 samples keep their keys in registers
                if (n <= KP * WAVE) {
                    unsigned long long keys[KP];
#pragma unroll
                    for (int q = 0; q < KP; q++) {
                        const int j = q * WAVE + lane;
                        keys[q] = j < n ? (unsigned long long)__double_as_longlong(dist[j]) : ~0ull;
                    }
                    while (lo < hi) {
                        const unsigned long long mid = lo + ((hi - lo) >> 1);
                        int cnt = 0;
#pragma unroll
                        for (int q = 0; q < KP; q++)
                            if (q * WAVE < n) cnt += __popcll(__ballot(keys[q] <= mid));
                        if (cnt == d.max_neighbours) { lo = hi = mid + 1; break; }   // exactly k keys <= mid: they are the k nearest
                        if (cnt >= d.max_neighbours) hi = mid; else lo = mid + 1;
                    }
                } else {
                    while (lo < hi) {
                        const unsigned long long mid = lo + ((hi - lo) >> 1);
                        int cnt = 0;
                        for (int j0 = 0; j0 < n; j0 += WAVE) {
                            const int j = j0 + lane;
                            const unsigned long long key = j < n ? (unsigned long long)__double_as_longlong(dist[j]) : ~0ull;
                            cnt += __popcll(__ballot(key <= mid));
                        }
                        if (cnt == d.max_neighbours) { lo = hi = mid + 1; break; }
                        if (cnt >= d.max_neighbours) hi = mid; else lo = mid + 1;
                    }
                }
                vstar = lo;
                for (int j0 = 0; j0 < n; j0 += WAVE) {
                    const int j = j0 + lane;
                    const unsigned long long key = j < n ? (unsigned long long)__double_as_longlong(dist[j]) : ~0ull;
                    n_less += __popcll(__ballot(key < vstar));
                }
            }
            int eq_taken = 0;
            for (int j0 = 0; j0 < n; j0 += WAVE) {
                const int j = j0 + lane;
                bool sel = false, eq = false;
                if (j < n) {
                    const double dj = dist[j];
                    if (top) {
                        const unsigned long long key = (unsigned long long)__double_as_longlong(dj);
                        sel = key < vstar;
                        eq = key == vstar;
                    } else
                        sel = dj < d.bandwidth;
                }
                if (top) {
                    const unsigned long long me = __ballot(eq);
                    const int r = eq_taken + __popcll(me & ((1ull << lane) - 1ull));
                    sel = sel || (eq && n_less + r < d.max_neighbours);
                    eq_taken += __popcll(me);
                }
                const unsigned long long m = __ballot(sel);
                const int pos = nsel + __popcll(m & ((1ull << lane) - 1ull));
                if (sel && nsel_tot + pos < 2 * LP_MAXNB) {
                    const double r = dist[j] / d.bandwidth;
                    isel[nsel_tot + pos] = (lapk << 16) | j;
                    wsel[nsel_tot + pos] = (1.0 - r * r) * 3.0 / 4.0;
                }
                nsel += __popcll(m);
            }
            nsel_tot += nsel;
            if (nsel_tot > 2 * LP_MAXNB) nsel_tot = 2 * LP_MAXNB;   // (max_neighbours <= 64 per lap; "inside" sets larger than that are the top-k case)
            SYNC();
        }
        // the 45 sums, one lane each: [0,15) Qa upper triangle, [15,30) Qd, [30,35) ba, [35,40) bd (vy), [40,45) bd (wz)
        {
            const int e = lane < 45 ? lane : 0;
            int kind, r2, c2;   // kind 0: Qa, 1: Qd, 2: ba, 3: bd0, 4: bd1
            if (e < 30) {
                kind = e / 15;
                int t = e - 15 * kind;
                r2 = 0;
                while (t >= 5 - r2) { t -= 5 - r2; r2++; }
                c2 = r2 + t;
            } else { kind = 2 + (e - 30) / 5; r2 = (e - 30) % 5; c2 = 0; }
            const bool use_a = kind == 0 || kind == 2;          // the vx row is driven by a (feature 4), vy / wz by delta (feature 3)
            // which two numbers of a sample's row this lane multiplies (offsets into [vx vy wz delta a | next vx vy wz]; -1 = the constant 1)
            const int fr = r2 < 3 ? r2 : (r2 == 3 ? (use_a ? 4 : 3) : -1);
            const int fc = kind < 2 ? (c2 < 3 ? c2 : (c2 == 3 ? (use_a ? 4 : 3) : -1)) : 5 + (kind - 2);
            double acc = 0.0;
            // four samples per step: their loads are independent (one at a time each sample cost two dependent LDS round trips),
            // the additions stay in sample order; a slot past the last sample adds +0.0, which changes nothing
            for (int q0 = 0; q0 < nsel_tot; q0 += 4) {
                double K[4], a[4], c[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + u < nsel_tot ? q0 + u : nsel_tot - 1;
                    const int pk = isel[q], lapk = pk >> 16, j = pk & 0xFFFF;
                    const double* F = feat + ((size_t)lapk * P + j) * 5;
                    K[u] = wsel[q];
                    a[u] = F[fr >= 0 ? fr : 0];
                    c[u] = F[fc >= 0 ? fc : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const double mr = fr >= 0 ? a[u] : 1.0, other = fc >= 0 ? c[u] : 1.0;
                    const double km = K[u] * mr;
                    const double t = km * other;
                    acc = acc + (q0 + u < nsel_tot ? t : 0.0);
                }
            }
            if (lane < 45) sums[lane] = acc;
        }
        SYNC();
        double Qa[5][5], Qd[5][5], ba[1][5], bd[2][5];
        {
            int e = 0;
            for (int r2 = 0; r2 < 5; r2++)
                for (int c2 = r2; c2 < 5; c2++, e++) { Qa[r2][c2] = Qa[c2][r2] = sums[e]; Qd[r2][c2] = Qd[c2][r2] = sums[15 + e]; }
            for (int r2 = 0; r2 < 5; r2++) { ba[0][r2] = sums[30 + r2]; bd[0][r2] = sums[35 + r2]; bd[1][r2] = sums[40 + r2]; }
        }
        const bool ok_a = lp_solve5<1>(Qa, ba), ok_d = lp_solve5<2>(Qd, bd);   // both are always solved (the oracle does)
        if (!ok_a || !ok_d) bad = 1;
        SYNC();
        if (lane == 0) {
            double* Ai = kp.A + ((size_t)b * N + i) * 36;
            double* Bi = kp.B + ((size_t)b * N + i) * 12;
            double* Ci = kp.C + ((size_t)b * N + i) * 6;
            // a singular normal matrix (no stored sample near this linearisation point; the reference's cvxopt raises) leaves
            // the three regression rows of the stage UNTOUCHED: a device-resident loop thereby keeps the previous step's
            // model of the stage, and status reports it
            if (ok_a && ok_d) {
                for (int k = 0; k < 18; k++) Ai[k] = 0.0;
                for (int k = 0; k < 6; k++) Bi[k] = 0.0;
                for (int k = 0; k < 3; k++) { Ai[0 * 6 + k] = ba[0][k]; Ai[1 * 6 + k] = bd[0][k]; Ai[2 * 6 + k] = bd[1][k]; }
                Bi[0 * 2 + 1] = ba[0][3]; Bi[1 * 2 + 0] = bd[0][3]; Bi[2 * 2 + 0] = bd[1][3];
                Ci[0] = ba[0][4]; Ci[1] = bd[0][4]; Ci[2] = bd[1][4];
            }
            for (int k = 6; k < 12; k++) Bi[k] = 0.0;
            // kinematic rows: analytic Jacobian of the Euler step (lmpc_helper.py:130-189, incl. `den * 2` at :163)
            const double vx = x0[0], vy = x0[1], wz = x0[2], epsi = x0[3], s = x0[4], ey = x0[5], dt = d.dt;
            const double cur = lp_curvature(kp.track, d.n_seg, d.lap_length, s);
            const double den = 1.0 - cur * ey, ce = cos(epsi), se = sin(epsi);
            const double along = vx * ce - vy * se, across = vx * se + vy * ce;
            double* r3 = Ai + 18; double* r4 = Ai + 24; double* r5 = Ai + 30;
            r3[0] = -dt * ce / den * cur; r3[1] = dt * se / den * cur; r3[2] = dt; r3[3] = 1.0 + dt * across / den * cur;
            r3[4] = 0.0; r3[5] = -dt * along / (den * den) * cur * cur;
            r4[0] = dt * ce / den; r4[1] = -dt * se / den; r4[2] = 0.0; r4[3] = -dt * across / den; r4[4] = 1.0;
            r4[5] = dt * along / (den * 2.0) * cur;
            r5[0] = dt * se; r5[1] = dt * ce; r5[2] = 0.0; r5[3] = dt * along; r5[4] = 0.0; r5[5] = 1.0;
            double d3 = 0.0, d4 = 0.0, d5 = 0.0;
            for (int k = 0; k < 6; k++) { d3 += r3[k] * x0[k]; d4 += r4[k] * x0[k]; d5 += r5[k] * x0[k]; }
            Ci[3] = epsi + dt * (wz - along / den * cur) - d3;
            Ci[4] = s + dt * along / den - d4;
            Ci[5] = ey + dt * across - d5;
        }
    }
    if (bad && lane == 0) atomicOr(&sbad, 1);
    // safe-set points: laps iter-1, iter-2 (control.py:625-639), n_ss_per_lap samples from `shift` after the 1-norm-nearest
    // (one lap per wave)
    const double* xb = kp.x + (size_t)b * 6;
    double xs[6];
    for (int k = 0; k < 6; k++) xs[k] = xb[k];
    for (int jj = wv; jj < d.n_ss_laps; jj += LP_WAVES) {
        const int lap = it - jj - 1;
        double best = INFINITY;
        int bj = 0x7fffffff;
        for (int j = lane; j < P; j += WAVE) {
            const double* sx = ss + ((size_t)lap * P + j) * 6;
            double s = 0.0;
            for (int k = 0; k < 6; k++) s = s + fabs(sx[k] - xs[k]);
            if (s < best) { best = s; bj = j; }                    // ascending j per lane: first minimum kept
        }
        const double wbest = wave_min(best);
        const int cand = best == wbest ? bj : 0x7fffffff;
        int first = cand;                                          // smallest index among the lanes holding the minimum
        for (int o = 32; o; o >>= 1) { const int other = __shfl_xor(first, o); first = other < first ? other : first; }
        const int lo = first + d.shift >= 0 ? first + d.shift : first;
        for (int e = lane; e < d.n_ss_per_lap * 7; e += WAVE) {
            const int q = e / 7, k = e - 7 * q;
            const int j = lo + q < P ? lo + q : P - 1;
            if (k < 6) kp.ss_sel[((size_t)b * 6 + k) * M + jj * d.n_ss_per_lap + q] = ss[((size_t)lap * P + j) * 6 + k];
            else kp.q_sel[(size_t)b * M + jj * d.n_ss_per_lap + q] = qf[(size_t)lap * P + j];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) kp.status[b] = sbad;
}

__global__ void __launch_bounds__(256) crx_lmpc_addpoint_kernel(const crx_lmpcprep_desc d, int batch, double* ss_xcurv, double* u_ss,
                                                                const int32_t* time_ss, const int32_t* iter, const int32_t* step,
                                                                const double* x, const double* u, int u_stride) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int P = d.n_points, L = d.n_laps;
    const int lap = iter[b] - 1;
    if (lap < 0 || lap >= L) return;
    const int row = time_ss[(size_t)b * L + lap] + step[b] + 1;
    if (row < 0 || row >= P) return;
    double* sx = ss_xcurv + (((size_t)b * L + lap) * P + row) * 6;
    for (int k = 0; k < 6; k++) sx[k] = x[(size_t)b * 6 + k] + (k == 4 ? d.lap_length : 0.0);
    double* su = u_ss + (((size_t)b * L + lap) * P + row) * 2;
    su[0] = u[(size_t)b * u_stride]; su[1] = u[(size_t)b * u_stride + 1];
}

// LMPCRacingGame.add_trajectory (utils/base.py:631-656) for the races that have just crossed the line: the logged lap becomes
// lap `iter` of the race's safe set (states, inputs, cost-to-go), the counters move on, the log restarts from the wrapped
// state.  One wavefront per race; the copies are coalesced, the cost-to-go recursion and the reference's second pass over
// the column run on one lane in the reference's order (oracle/crx_oracle_lmpc_prep.c crx_oracle_lmpc_addtraj).
__global__ void __launch_bounds__(WAVE) crx_lmpc_addtraj_kernel(const crx_lmpcprep_desc d, int batch, const int32_t* crossed, double* log_x,
                                                                const double* log_u, int32_t* n_log, double* ss_xcurv, double* u_ss,
                                                                double* qfun, int32_t* time_ss, int32_t* iter, int32_t* step,
                                                                const double* x, int32_t* status) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= batch) return;
    if (!crossed[b]) {
        if (lane == 0) status[b] = 0;
        return;
    }
    const int P = d.n_points, L = d.n_laps, lap = iter[b];
    int n = n_log[b] - 1;
    n = n > P - 1 ? P - 1 : n;
    double* lx = log_x + (size_t)b * P * 6;
    const bool room = lap >= 0 && lap < L && n >= 1;
    if (room) {
        double* sx = ss_xcurv + ((size_t)b * L + lap) * P * 6;
        double* su = u_ss + ((size_t)b * L + lap) * P * 2;
        const double* lu = log_u + (size_t)b * P * 2;
        for (int e = lane; e < (n + 1) * 6; e += WAVE) sx[e] = lx[e];
        for (int e = lane; e < n * 2; e += WAVE) su[e] = lu[e];
        if (lane == 0) {
            double* q = qfun + ((size_t)b * L + lap) * P;
            q[n] = 0.0;
            for (int i = n - 1; i >= 0; i--) q[i] = lx[6 * i + 4] < d.lap_length ? q[i + 1] + 1.0 : 0.0;
            for (int i = 0; i < P; i++)
                if (q[i] == 0.0) q[i] = q[i > 0 ? i - 1 : P - 1] - 1.0;
            time_ss[(size_t)b * L + lap] = n;
        }
    }
    SYNC();   // (the copies above read lx[0..5]; a single wave: program order suffices)
    if (lane < 6) lx[lane] = x[(size_t)b * 6 + lane];
    if (lane == 0) {
        if (room) iter[b] = lap + 1;
        status[b] = room ? 0 : 1;
        step[b] = 0;
        n_log[b] = 1;
    }
}

size_t crx_lmpcprep_lds_bytes(int n_points) {
    return ((size_t)2 * n_points * 5 + LP_WAVES * LP_WAVE_DOUBLES(n_points)) * sizeof(double);
}

hipError_t crx_launch_lmpcprep(const crx_lmpcprep_kparams& kp, hipStream_t st) {
    if (kp.batch == 0) return hipSuccess;
    const size_t bytes = crx_lmpcprep_lds_bytes(kp.d.n_points);
    static int attr_set_on = -1;
    static size_t attr_bytes = 0;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (attr_set_on != dev || bytes > attr_bytes) {
        hipError_t e = hipFuncSetAttribute((const void*)crx_lmpc_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        attr_set_on = dev; attr_bytes = bytes;
    }
    hipLaunchKernelGGL(crx_lmpc_prep_kernel, dim3(kp.batch), dim3(WAVE * LP_WAVES), bytes, st, kp);
    return hipGetLastError();
}

hipError_t crx_launch_lmpc_addpoint(const crx_lmpcprep_desc& d, int batch, double* ss_xcurv, double* u_ss, const int32_t* time_ss,
                                    const int32_t* iter, const int32_t* step, const double* x, const double* u, int u_stride,
                                    hipStream_t st) {
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_lmpc_addpoint_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, d, batch, ss_xcurv, u_ss, time_ss, iter,
                       step, x, u, u_stride);
    return hipGetLastError();
}

hipError_t crx_launch_lmpc_addtraj(const crx_lmpcprep_desc& d, int batch, const int32_t* crossed, double* log_x, const double* log_u,
                                   int32_t* n_log, double* ss_xcurv, double* u_ss, double* qfun, int32_t* time_ss, int32_t* iter,
                                   int32_t* step, const double* x, int32_t* status, hipStream_t st) {
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_lmpc_addtraj_kernel, dim3(batch), dim3(WAVE), 0, st, d, batch, crossed, log_x, log_u, n_log, ss_xcurv, u_ss,
                       qfun, time_ss, iter, step, x, status);
    return hipGetLastError();
}
