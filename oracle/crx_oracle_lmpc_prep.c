/*
 * crx_oracle_lmpc_prep.c -- CPU restatement (plain C, double) of the host work in front of the learning-MPC QP:
 * the local LTV model regression, the kinematic linearisation and the safe-set point selection.
 *
 * TEST INFRASTRUCTURE (see crx_oracle.c): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it.
 *
 * What is restated (paths into /root/reference/car_racing):
 *   control/lmpc_helper.py:26-189   regression_and_linearization -- one stage model (A_i, B_i, C_i)
 *                      :192-226     compute_index   neighbours within the bandwidth h = 5 of the scaled 1-norm, at most
 *                                   max_num_point nearest, Epanechnikov weights (1 - (d/h)^2) 3/4
 *                      :229-275     compute_Q_M     normal matrix M'KM over [vx vy wz u_f 1], lamb = 0
 *                      :338-355     compute_b       right-hand side M'K y for the next-step vx / vy / wz
 *                      :358-366     lmpc_loc_lin_reg  cvxopt qp(Q, b) WITHOUT constraints = solve Q w = -b
 *                      :278-293     select_points   num_ss_points/num_ss_iter points after the 1-norm-nearest sample
 *   utils/base.py:585-622           estimate_ABC    the loop over the N stages, used_iter = (iter-2, iter-1)
 *   control/control.py:625-639      lmpc            safe-set selection over the laps iter-1, iter-2
 *   utils/racing_env.py:225-246     get_curvature
 *
 * Arithmetic order.  The reference forms M'KM with BLAS and solves with cvxopt/LAPACK; the normal matrix has a condition
 * number up to 3e11 (vy ~ 0.1 wz along the whole data set), so individual coefficients are determined to ~1e-5 relative
 * between ANY two routes.  This file fixes one order -- samples in ascending index order lap by lap, plain
 * multiply-add without fused operations, Gaussian elimination with partial pivoting -- and crx_lmpc_prep_kernel follows
 * the same order operation by operation, so kernel and oracle agree to the last bits of the solve although both are
 * only ~1e-5 from the reference's own coefficients (tests/test_oracle_golden.py pins predictions at the query point
 * tightly, coefficients loosely, like tests/test_host_mirror.py does for the numpy mirror).
 *
 * Safe-set layout here and in libcrx: [race][lap][point][component] (the reference stores [point][component][lap]).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/crx.h"

#define PMAX 4096

/* racing_env.get_curvature: s wrapped into one lap, first segment with lo <= s <= hi */
static double curvature(const double* track, int n_seg, double lap_length, double s) {
    while (s > lap_length) s -= lap_length;
    while (s < 0.0) s += lap_length;
    for (int i = 0; i < n_seg; i++)
        if (s >= track[6 * i + 3] && s <= track[6 * i + 3] + track[6 * i + 4]) return track[6 * i + 5];
    return 0.0;
}

/* Gaussian elimination with partial pivoting on a 5x5 system with nrhs right-hand sides (no fused operations).
 * Returns 0 if a pivot is exactly zero or not finite (the reference's cvxopt raises on a singular matrix). */
static int solve5(double Q[5][5], double rhs[][5], int nrhs) {
    for (int c = 0; c < 5; c++) {
        int p = c;
        double best = fabs(Q[c][c]);
        for (int r = c + 1; r < 5; r++)
            if (fabs(Q[r][c]) > best) { best = fabs(Q[r][c]); p = r; }
        if (!(best > 0.0) || !isfinite(best)) return 0;
        if (p != c) {
            for (int k = 0; k < 5; k++) { double t = Q[c][k]; Q[c][k] = Q[p][k]; Q[p][k] = t; }
            for (int q = 0; q < nrhs; q++) { double t = rhs[q][c]; rhs[q][c] = rhs[q][p]; rhs[q][p] = t; }
        }
        for (int r = c + 1; r < 5; r++) {
            const double f = Q[r][c] / Q[c][c];
            for (int k = c; k < 5; k++) { const double t = f * Q[c][k]; Q[r][k] = Q[r][k] - t; }
            for (int q = 0; q < nrhs; q++) { const double t = f * rhs[q][c]; rhs[q][r] = rhs[q][r] - t; }
        }
    }
    for (int q = 0; q < nrhs; q++)
        for (int r = 4; r >= 0; r--) {
            double s = rhs[q][r];
            for (int k = r + 1; k < 5; k++) { const double t = Q[r][k] * rhs[q][k]; s = s - t; }
            rhs[q][r] = s / Q[r][r];
        }
    return 1;
}

int crx_oracle_lmpc_prep(const crx_lmpcprep_desc* d, int batch, const double* ss_xcurv, const double* u_ss,
                         const double* qfun, const int32_t* time_ss, const int32_t* iter, const double* x,
                         const double* lin_points, const double* lin_input, int from_plan, const double* track,
                         double* A, double* B, double* C, double* ss_sel, double* q_sel, int32_t* status) {
    if (!d || d->N < 2 || d->N > CRX_LMPC_MAX_N || d->n_points < 2 || d->n_points > PMAX || d->n_laps < 2 || batch < 0 ||
        d->n_ss_laps < 1 || d->n_ss_laps > 2 || d->n_ss_per_lap < 1 || d->max_neighbours < 1 || d->max_neighbours > 64)
        return CRX_ERR_ARG;
    const int N = d->N, P = d->n_points, L = d->n_laps, M = d->n_ss_per_lap * d->n_ss_laps;
    static _Thread_local double dist[PMAX];
    static _Thread_local int order[PMAX];
    for (int b = 0; b < batch; b++) {
        const double* ss = ss_xcurv + (size_t)b * L * P * 6;
        const double* us = u_ss + (size_t)b * L * P * 2;
        const double* qf = qfun + (size_t)b * L * P;
        const int it = iter[b];
        status[b] = 0;
        if (it < 2 || it > L) return CRX_ERR_ARG;
        for (int i = 0; i < N; i++) {
            /* linearisation point: given, or the previous plan shifted by one stage (control.py:726-728) */
            double x0[6], u0[2];
            const int ix = from_plan ? (i + 1 <= N ? i + 1 : N) : i, iu = from_plan ? (i + 1 <= N - 1 ? i + 1 : N - 1) : i;
            memcpy(x0, lin_points + ((size_t)b * (N + 1) + ix) * 6, sizeof(x0));
            memcpy(u0, lin_input + ((size_t)b * N + iu) * 2, sizeof(u0));
            const double xl[5] = {x0[0], x0[1], x0[2], u0[0], u0[1]};
            /* normal equations, samples in ascending index order, lap iter-2 first (used_iter = range(iter-2, iter)) */
            double Qa[5][5] = {{0}}, Qd[5][5] = {{0}}, ba[1][5] = {{0}}, bd[2][5] = {{0}};
            for (int lapk = 0; lapk < 2; lapk++) {
                const int lap = it - 2 + lapk;
                const int n = time_ss[(size_t)b * L + lap] - 1;
                if (n < 1 || n + 1 > P) return CRX_ERR_ARG;
                int inside = 0;
                for (int j = 0; j < n; j++) {
                    const double* sx = ss + ((size_t)lap * P + j) * 6;
                    const double* su = us + ((size_t)lap * P + j) * 2;
                    double s = 0.0;
                    s += fabs((sx[0] - xl[0]) * d->scale[0]);
                    s += fabs((sx[1] - xl[1]) * d->scale[1]);
                    s += fabs((sx[2] - xl[2]) * d->scale[2]);
                    s += fabs((su[0] - xl[3]) * d->scale[3]);
                    s += fabs((su[1] - xl[4]) * d->scale[4]);
                    dist[j] = s;
                    if (s < d->bandwidth) inside++;
                }
                /* selected set: everything inside the bandwidth, or the max_neighbours nearest (rank by (dist, index)) */
                int nsel = 0;
                for (int j = 0; j < n; j++) {
                    int sel;
                    if (inside >= d->max_neighbours) {
                        int rank = 0;
                        for (int k = 0; k < n; k++) rank += (dist[k] < dist[j]) || (dist[k] == dist[j] && k < j);
                        sel = rank < d->max_neighbours;
                    } else
                        sel = dist[j] < d->bandwidth;
                    if (sel) order[nsel++] = j;
                }
                for (int q = 0; q < nsel; q++) {
                    const int j = order[q];
                    const double* sx = ss + ((size_t)lap * P + j) * 6;
                    const double* su = us + ((size_t)lap * P + j) * 2;
                    const double* sn = ss + ((size_t)lap * P + j + 1) * 6;
                    const double r = dist[j] / d->bandwidth;
                    const double K = (1.0 - r * r) * 3.0 / 4.0;
                    const double ma[5] = {sx[0], sx[1], sx[2], su[1], 1.0};   /* vx row: driven by a      (input_features = [1]) */
                    const double md[5] = {sx[0], sx[1], sx[2], su[0], 1.0};   /* vy, wz rows: by delta    (input_features = [0]) */
                    for (int r2 = 0; r2 < 5; r2++) {
                        const double ka = K * ma[r2], kd = K * md[r2];
                        for (int c2 = r2; c2 < 5; c2++) {              /* upper triangle; mirrored below */
                            const double ta = ka * ma[c2], td = kd * md[c2];
                            Qa[r2][c2] = Qa[r2][c2] + ta;
                            Qd[r2][c2] = Qd[r2][c2] + td;
                        }
                        const double t0 = ka * sn[0], t1 = kd * sn[1], t2 = kd * sn[2];
                        ba[0][r2] = ba[0][r2] + t0;
                        bd[0][r2] = bd[0][r2] + t1;
                        bd[1][r2] = bd[1][r2] + t2;
                    }
                }
            }
            for (int r2 = 0; r2 < 5; r2++)
                for (int c2 = 0; c2 < r2; c2++) { Qa[r2][c2] = Qa[c2][r2]; Qd[r2][c2] = Qd[c2][r2]; }
            double* Ai = A + ((size_t)b * N + i) * 36;
            double* Bi = B + ((size_t)b * N + i) * 12;
            double* Ci = C + ((size_t)b * N + i) * 6;
            const int ok = solve5(Qa, ba, 1) & solve5(Qd, bd, 2);
            if (!ok) status[b] = 1;   /* singular normal matrix: the reference's cvxopt raises (lmpc_helper.py:358-366); the
                                         regression rows of the stage are left untouched (see the kernel) */
            else {
                memset(Ai, 0, sizeof(double) * 18); memset(Bi, 0, sizeof(double) * 6);
                for (int k = 0; k < 3; k++) { Ai[0 * 6 + k] = ba[0][k]; Ai[1 * 6 + k] = bd[0][k]; Ai[2 * 6 + k] = bd[1][k]; }
                Bi[0 * 2 + 1] = ba[0][3]; Bi[1 * 2 + 0] = bd[0][3]; Bi[2 * 2 + 0] = bd[1][3];
                Ci[0] = ba[0][4]; Ci[1] = bd[0][4]; Ci[2] = bd[1][4];
            }
            memset(Bi + 6, 0, sizeof(double) * 6);
            /* kinematic rows: analytic Jacobian of the Euler step (lmpc_helper.py:130-189, incl. `den * 2` at :163) */
            const double vx = x0[0], vy = x0[1], wz = x0[2], epsi = x0[3], s = x0[4], ey = x0[5], dt = d->dt;
            const double cur = curvature(track, d->n_seg, d->lap_length, s);
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
        /* safe-set points: laps iter-1, iter-2 (control.py:625-639), n_ss_per_lap samples from `shift` after the nearest */
        const double* xb = x + (size_t)b * 6;
        for (int jj = 0; jj < d->n_ss_laps; jj++) {
            const int lap = it - jj - 1;
            int best = 0;
            double bd2 = HUGE_VAL;
            for (int j = 0; j < P; j++) {
                const double* sx = ss + ((size_t)lap * P + j) * 6;
                double s = 0.0;
                for (int k = 0; k < 6; k++) s += fabs(sx[k] - xb[k]);
                if (s < bd2) { bd2 = s; best = j; }      /* np.argmin: first minimum */
            }
            int lo = best + d->shift >= 0 ? best + d->shift : best;
            for (int q = 0; q < d->n_ss_per_lap; q++) {
                const int j = lo + q < P ? lo + q : P - 1;   /* the reference would return a shorter slice past the end of its arrays */
                for (int k = 0; k < 6; k++) ss_sel[((size_t)b * 6 + k) * M + jj * d->n_ss_per_lap + q] = ss[((size_t)lap * P + j) * 6 + k];
                q_sel[(size_t)b * M + jj * d->n_ss_per_lap + q] = qf[(size_t)lap * P + j];
            }
        }
    }
    return CRX_OK;
}

/* LMPCRacingGame.add_point (utils/base.py:624-629): extend lap iter-1 past the finish line with the running lap */
int crx_oracle_lmpc_addpoint(const crx_lmpcprep_desc* d, int batch, double* ss_xcurv, double* u_ss, const int32_t* time_ss,
                             const int32_t* iter, const int32_t* step, const double* x, const double* u, int u_stride) {
    if (!d || batch < 0 || u_stride < 2) return CRX_ERR_ARG;
    const int P = d->n_points, L = d->n_laps;
    for (int b = 0; b < batch; b++) {
        const int lap = iter[b] - 1;
        if (lap < 0 || lap >= L) return CRX_ERR_ARG;
        const int row = time_ss[(size_t)b * L + lap] + step[b] + 1;
        if (row < 0 || row >= P) continue;   /* outside the arrays (the reference would raise IndexError) */
        double* sx = ss_xcurv + (((size_t)b * L + lap) * P + row) * 6;
        for (int k = 0; k < 6; k++) sx[k] = x[(size_t)b * 6 + k] + (k == 4 ? d->lap_length : 0.0);
        double* su = u_ss + (((size_t)b * L + lap) * P + row) * 2;
        su[0] = u[(size_t)b * u_stride]; su[1] = u[(size_t)b * u_stride + 1];
    }
    return CRX_OK;
}

/* LMPCRacingGame.add_trajectory (utils/base.py:631-656): the lap a race has just completed becomes lap `iter` of its safe set.
 * log_x [batch][P][6] / log_u [batch][P][2] hold the running lap as the simulator logs it (update_memory, base.py:795-819):
 * n_log states, the last one the crossing state with s > lap_length (unwrapped), n_log - 1 inputs.  For every race with
 * crossed != 0:  ss[lap][0..n] = states, us[lap][0..n-1] = inputs, time_ss[lap] = n = n_log - 1,  Qfun = compute_cost
 * (lmpc_helper.py:11-23: 0 at the last sample and past the finish line, else one more than the successor) followed by the
 * reference's "zero means keep counting down" pass over the WHOLE column (:647-649; entry 0 would read entry -1, i.e. the
 * last one, like numpy does),  iter += 1, step = 0 (time_in_iter), and the log restarts with x (the wrapped state the new lap
 * starts from).  A race whose safe set is full (iter == n_laps) keeps racing without storing: status 1. */
int crx_oracle_lmpc_addtraj(const crx_lmpcprep_desc* d, int batch, const int32_t* crossed, double* log_x, const double* log_u,
                            int32_t* n_log, double* ss_xcurv, double* u_ss, double* qfun, int32_t* time_ss, int32_t* iter,
                            int32_t* step, const double* x, int32_t* status) {
    if (!d || batch < 0) return CRX_ERR_ARG;
    const int P = d->n_points, L = d->n_laps;
    for (int b = 0; b < batch; b++) {
        status[b] = 0;
        if (!crossed[b]) continue;
        const int lap = iter[b];
        int n = n_log[b] - 1;
        if (n > P - 1) n = P - 1;
        double* lx = log_x + (size_t)b * P * 6;
        if (lap >= 0 && lap < L && n >= 1) {
            double* sx = ss_xcurv + ((size_t)b * L + lap) * P * 6;
            double* su = u_ss + ((size_t)b * L + lap) * P * 2;
            double* q = qfun + ((size_t)b * L + lap) * P;
            const double* lu = log_u + (size_t)b * P * 2;
            for (int e = 0; e < (n + 1) * 6; e++) sx[e] = lx[e];
            for (int e = 0; e < n * 2; e++) su[e] = lu[e];
            time_ss[(size_t)b * L + lap] = n;
            q[n] = 0.0;
            for (int i = n - 1; i >= 0; i--) q[i] = lx[6 * i + 4] < d->lap_length ? q[i + 1] + 1.0 : 0.0;
            for (int i = 0; i < P; i++)
                if (q[i] == 0.0) q[i] = q[i > 0 ? i - 1 : P - 1] - 1.0;
            iter[b] = lap + 1;
        } else
            status[b] = 1;
        step[b] = 0;
        for (int k = 0; k < 6; k++) lx[k] = x[(size_t)b * 6 + k];
        n_log[b] = 1;
    }
    return CRX_OK;
}
