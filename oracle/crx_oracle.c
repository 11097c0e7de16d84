/*
 * crx_oracle.c -- CPU restatement (plain C, double) of the car-racing planner / MPC-CBF hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this; the product (libcrx, HIP) never links, imports or falls back to it.
 *
 * What is restated, and from where (paths into /root/reference/car_racing):
 *   problem construction
 *     planner region QP     planning/overtake_traj_planner.py:263-334  (cost :325-334, bounds :276-284,
 *                           dynamics :270-274, reference points :329-332, fall-back :365-374)
 *     MPC-CBF NLP           control/control.py:492-591 (mpccbf) and :270-382 (mpc_multi_agents):
 *                           CBF rows :537-562, dynamics :566-570, input box :572-576,
 *                           state box + tracking cost :580-591, input cost :578-579
 *     region selection      planning/overtake_traj_planner.py:205-246
 *     linear interpolation  scipy.interpolate.interp1d(kind="linear") as used at
 *                           planning/overtake_traj_planner.py:112-117,332
 *   solver
 *     The reference delegates the arithmetic to casadi==3.5.5 -> IPOPT (requirements.txt:6;
 *     call sites control.py:593-599, overtake_traj_planner.py:359-364), a third-party wheel that is
 *     not in the reference tree and cannot be installed here.  What follows restates IPOPT's
 *     published algorithm (Waechter & Biegler, Math. Prog. 106, 2006): slack form g(x)-t=0, t>=0
 *     for every inequality (CasADi Opti passes all constraints as general g), monotone
 *     Fiacco-McCormick barrier update (eq. 7), fraction-to-the-boundary rule (eq. 15), primal-dual
 *     Newton step with inertia correction by W + delta_w*I (Alg. IC), multiplier safeguard (eq. 16),
 *     gradient-based constraint scaling (sec. 3.8), error measure E_mu (eq. 5), filter line search
 *     (sec. 2.3) without second-order correction; the restoration phase is the closed-form one of
 *     restore_slacks() (CBF NLP only).
 *     Because the dynamics are linear and x0 is fixed, states are eliminated (condensing) and the
 *     reduced Newton system is factorised by a DENSE Cholesky -- on purpose a different linear-algebra
 *     route from the HIP kernel's Riccati recursion, so that agreement between the two is evidence.
 *
 * PARITY PIN: the reference's own tests hold no golden vectors for this path.
 * This oracle is pinned instead against tests/golden/ *.npz: problems recorded from the reference's
 * own, unmodified code running against a recording CasADi stand-in, with independently certified
 * KKT solutions (tests/golden/tools/make_golden.py).  IPOPT itself never ran: "parity vs IPOPT's
 * iterates" is unpinned; "parity vs the reference's problem + a certified KKT point" is pinned.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/crx.h"

#define MAXN CRX_MAX_N
#define MAXO CRX_MAX_OBS
#define NXS 6
#define NUS 2
#define MAXRED (MAXN * (NUS + MAXO) + MAXO)
#define MAXM (MAXN * 4 + MAXN * 4 + MAXO * (MAXN + 1) + MAXO * MAXN)

/* ------------------------------------------------------------------------------------------------
 * canonical stage-structured problem shared by both front-ends
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int N, nobs;
    double A[36], B[12];
    double x0[6];
    double wq[6];                 /* cost  sum_k sum_i wq_i (x_ki - xr_ki)^2           k = 0..N  */
    double xr[MAXN + 1][6];
    double lin[MAXN + 1][6];      /*     + sum_k lin_k . x_k + cconst                            */
    double cconst;
    double wr[2];                 /*     + sum_k sum_i wr_i u_ki^2                      k = 0..N-1 */
    double wc[MAXN];              /*     + sum_k wc_k (ey_{k+1} - ey_k)^2               k = 0..N-1 */
    double wsig;                  /*     + wsig * sum sigma                                      */
    double ulo[2], uhi[2];
    double vlo[MAXN + 1], vhi[MAXN + 1], elo[MAXN + 1], ehi[MAXN + 1]; /* +-HUGE_VAL = absent */
    double obs_s[MAXO][MAXN + 1], obs_ey[MAXO][MAXN + 1], lap_off[MAXO];
    double alpha, cm, Ls[MAXO], Ws[MAXO];   /* l_agent + l_obs, w_agent + w_obs per obstacle (control.py:529-535) */
    int degree;
    int linear_rows;              /* every row is linear in v (no obstacle slot in the descriptor): box_certificate() applies */
} ocp_t;

enum { ROW_ULO, ROW_UHI, ROW_XLO, ROW_XHI, ROW_SIG, ROW_CBF };

typedef struct {
    int kind, k, i, o; /* stage, component (u index or state index), obstacle */
} rowdef_t;

typedef struct {
    const ocp_t* p;
    const crx_ipm_opts* o;
    int nred, m, nsv; /* nsv = 2 + nobs inputs per stage */
    rowdef_t row[MAXM];
    double d[MAXM];                       /* row scaling */
    double Sx[MAXN + 1][6][MAXRED];       /* state sensitivities dx_k/dv */
    double Hf[MAXRED][MAXRED];            /* constant cost Hessian in v */
    /* iterate */
    double v[MAXRED], t[MAXM], nu[MAXM];
    double x[MAXN + 1][6];
    double u[MAXN][2];
    double sig[MAXO][MAXN + 1];
    /* work */
    double f, g[MAXRED], c[MAXM], J[MAXM][MAXRED];
    double H[MAXRED][MAXRED], rhs[MAXRED], dv[MAXRED], dt[MAXM], dnu[MAXM];
} work_t;

static inline int iu(const work_t* w, int k) { return (w->p->N - 1 - k) * w->nsv; }
static inline int isig(const work_t* w, int k, int o) {
    return k == 0 ? w->p->N * w->nsv + o : iu(w, k - 1) + 2 + o;
}

static inline double ipow(double a, int p) {
    double r = 1.0;
    for (int i = 0; i < p; i++) r *= a;
    return r;
}

/* v -> (u, sigma, x) */
static void unpack(work_t* w, const double* v) {
    const ocp_t* p = w->p;
    for (int k = 0; k < p->N; k++) {
        w->u[k][0] = v[iu(w, k)];
        w->u[k][1] = v[iu(w, k) + 1];
    }
    for (int o = 0; o < p->nobs; o++)
        for (int k = 0; k <= p->N; k++) w->sig[o][k] = v[isig(w, k, o)];
    memcpy(w->x[0], p->x0, sizeof(double) * 6);
    for (int k = 0; k < p->N; k++)
        for (int i = 0; i < 6; i++) {
            double s = 0.0;
            for (int j = 0; j < 6; j++) s += p->A[i * 6 + j] * w->x[k][j];
            s += p->B[i * 2] * w->u[k][0] + p->B[i * 2 + 1] * w->u[k][1];
            w->x[k + 1][i] = s;
        }
}

static double cost_value(const work_t* w) {
    const ocp_t* p = w->p;
    double f = p->cconst;
    for (int k = 0; k <= p->N; k++)
        for (int i = 0; i < 6; i++) {
            double e = w->x[k][i] - p->xr[k][i];
            f += p->wq[i] * e * e + p->lin[k][i] * w->x[k][i];
        }
    for (int k = 0; k < p->N; k++) {
        f += p->wr[0] * w->u[k][0] * w->u[k][0] + p->wr[1] * w->u[k][1] * w->u[k][1];
        double de = w->x[k + 1][5] - w->x[k][5];
        f += p->wc[k] * de * de;
    }
    for (int o = 0; o < p->nobs; o++)
        for (int k = 0; k <= p->N; k++) f += p->wsig * w->sig[o][k];
    return f;
}

/* CBF pieces for obstacle o between stages i and i+1 (control.py:537-558):
 *   c = g_next(x_{i+1}) - sigma_{i+1} - (1-alpha) (g_cur(x_i) - sigma_i) - alpha*(1+margin)      */
static inline void cbf_terms(const ocp_t* p, const double x[][6], int o, int i, double* dsc,
                             double* dec, double* dsn, double* den) {
    *dsc = (x[i][4] - p->obs_s[o][i] - p->lap_off[o]) / p->Ls[o];     /* lap-corrected (:539-540) */
    *dec = (x[i][5] - p->obs_ey[o][i]) / p->Ws[o];
    *dsn = (x[i + 1][4] - p->obs_s[o][i + 1]) / p->Ls[o];             /* NOT lap-corrected (:542) */
    *den = (x[i + 1][5] - p->obs_ey[o][i + 1]) / p->Ws[o];
}

static double row_value(const work_t* w, int j) {
    const ocp_t* p = w->p;
    const rowdef_t* r = &w->row[j];
    switch (r->kind) {
        case ROW_ULO: return w->u[r->k][r->i] - p->ulo[r->i];
        case ROW_UHI: return p->uhi[r->i] - w->u[r->k][r->i];
        case ROW_XLO: return w->x[r->k][r->i] - (r->i == 0 ? p->vlo[r->k] : p->elo[r->k]);
        case ROW_XHI: return (r->i == 0 ? p->vhi[r->k] : p->ehi[r->k]) - w->x[r->k][r->i];
        case ROW_SIG: return w->sig[r->o][r->k];
        default: {
            double dsc, dec, dsn, den;
            cbf_terms(p, w->x, r->o, r->k, &dsc, &dec, &dsn, &den);
            int q = p->degree;
            double gc = ipow(dsc, q) + ipow(dec, q), gn = ipow(dsn, q) + ipow(den, q);
            return gn - w->sig[r->o][r->k + 1] - (1.0 - p->alpha) * (gc - w->sig[r->o][r->k]) -
                   p->alpha * p->cm;
        }
    }
}

/* dense Jacobian row in v (unscaled) */
static void row_jac(const work_t* w, int j, double* out) {
    const ocp_t* p = w->p;
    const rowdef_t* r = &w->row[j];
    int n = w->nred;
    memset(out, 0, sizeof(double) * n);
    switch (r->kind) {
        case ROW_ULO: out[iu(w, r->k) + r->i] = 1.0; break;
        case ROW_UHI: out[iu(w, r->k) + r->i] = -1.0; break;
        case ROW_XLO:
            for (int a = 0; a < n; a++) out[a] = w->Sx[r->k][r->i][a];
            break;
        case ROW_XHI:
            for (int a = 0; a < n; a++) out[a] = -w->Sx[r->k][r->i][a];
            break;
        case ROW_SIG: out[isig(w, r->k, r->o)] = 1.0; break;
        default: {
            double dsc, dec, dsn, den;
            int i = r->k, q = p->degree;
            cbf_terms(p, w->x, r->o, i, &dsc, &dec, &dsn, &den);
            double gsn = q * ipow(dsn, q - 1) / p->Ls[r->o], gen = q * ipow(den, q - 1) / p->Ws[r->o];
            double gsc = q * ipow(dsc, q - 1) / p->Ls[r->o], gec = q * ipow(dec, q - 1) / p->Ws[r->o];
            double om = 1.0 - p->alpha;
            for (int a = 0; a < n; a++)
                out[a] = gsn * w->Sx[i + 1][4][a] + gen * w->Sx[i + 1][5][a] -
                         om * (gsc * w->Sx[i][4][a] + gec * w->Sx[i][5][a]);
            out[isig(w, i + 1, r->o)] -= 1.0;
            out[isig(w, i, r->o)] += om;
        }
    }
}

/* full evaluation at w->v: x,u,sigma, f, g, c (scaled), J (scaled) */
static void eval_full(work_t* w) {
    const ocp_t* p = w->p;
    int n = w->nred, N = p->N;
    unpack(w, w->v);
    w->f = cost_value(w);
    /* gradient: adjoint sweep */
    double lam[6] = {0, 0, 0, 0, 0, 0};
    memset(w->g, 0, sizeof(double) * n);
    for (int k = N; k >= 1; k--) {
        double gx[6];
        for (int i = 0; i < 6; i++)
            gx[i] = 2.0 * p->wq[i] * (w->x[k][i] - p->xr[k][i]) + p->lin[k][i] + lam[i];
        /* coupling terms touching ey_k: wc_{k-1}(ey_k - ey_{k-1})^2 and wc_k(ey_{k+1}-ey_k)^2 */
        gx[5] += 2.0 * p->wc[k - 1] * (w->x[k][5] - w->x[k - 1][5]);
        if (k < N) gx[5] -= 2.0 * p->wc[k] * (w->x[k + 1][5] - w->x[k][5]);
        /* gx now = dL/dx_k including costate of later stages; push through x_k = A x_{k-1} + B u_{k-1} */
        for (int c = 0; c < 2; c++) {
            double s = 0.0;
            for (int i = 0; i < 6; i++) s += p->B[i * 2 + c] * gx[i];
            w->g[iu(w, k - 1) + c] = s + 2.0 * p->wr[c] * w->u[k - 1][c];
        }
        for (int j = 0; j < 6; j++) {
            double s = 0.0;
            for (int i = 0; i < 6; i++) s += p->A[i * 6 + j] * gx[i];
            lam[j] = s;
        }
    }
    for (int o = 0; o < p->nobs; o++)
        for (int k = 0; k <= N; k++) w->g[isig(w, k, o)] = p->wsig;
    for (int j = 0; j < w->m; j++) {
        w->c[j] = w->d[j] * row_value(w, j);
        if (w->row[j].kind == ROW_CBF || w->J[j][0] != w->J[j][0] /* first fill */) {
            row_jac(w, j, w->J[j]);
            for (int a = 0; a < n; a++) w->J[j][a] *= w->d[j];
        }
    }
}

/* cheap evaluation for the line search: f and c at a trial v */
static void eval_fc(work_t* w, const double* v, double* f, double* c) {
    unpack(w, v);
    *f = cost_value(w);
    for (int j = 0; j < w->m; j++) c[j] = w->d[j] * row_value(w, j);
}

static long g_nchol = 0, g_niter = 0, g_nsoc = 0, g_nwatch = 0;
long crx_oracle_stat(int i) { long v = i == 3 ? g_nwatch : (i == 2 ? g_nsoc : (i ? g_niter : g_nchol)); if (i < 0) { g_nchol = g_niter = g_nsoc = g_nwatch = 0; } return v; }
static int chol(int n, double H[][MAXRED]) {
#pragma omp atomic
    g_nchol++;
    for (int j = 0; j < n; j++) {
        double s = H[j][j];
        for (int k = 0; k < j; k++) s -= H[j][k] * H[j][k];
        if (!(s > 0.0)) return 0;
        double l = sqrt(s);
        H[j][j] = l;
        for (int i = j + 1; i < n; i++) {
            double t = H[i][j];
            for (int k = 0; k < j; k++) t -= H[i][k] * H[j][k];
            H[i][j] = t / l;
        }
    }
    return 1;
}

static void chol_solve(int n, double L[][MAXRED], double* b) {
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i][k] * b[k];
        b[i] = s / L[i][i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L[k][i] * b[k];
        b[i] = s / L[i][i];
    }
}

static void setup(work_t* w, const ocp_t* p, const crx_ipm_opts* o) {
    int N = p->N;
    w->p = p;
    w->o = o;
    w->nsv = 2 + p->nobs;
    w->nred = N * w->nsv + p->nobs;
    int n = w->nred;
    /* sensitivities */
    for (int k = 0; k <= N; k++)
        for (int i = 0; i < 6; i++) memset(w->Sx[k][i], 0, sizeof(double) * n);
    for (int k = 0; k < N; k++)
        for (int i = 0; i < 6; i++) {
            for (int a = 0; a < n; a++) {
                double s = 0.0;
                for (int j = 0; j < 6; j++) s += p->A[i * 6 + j] * w->Sx[k][j][a];
                w->Sx[k + 1][i][a] = s;
            }
            w->Sx[k + 1][i][iu(w, k)] += p->B[i * 2];
            w->Sx[k + 1][i][iu(w, k) + 1] += p->B[i * 2 + 1];
        }
    /* constant cost Hessian */
    for (int a = 0; a < n; a++) memset(w->Hf[a], 0, sizeof(double) * n);
    for (int k = 1; k <= N; k++) {
        for (int i = 0; i < 6; i++) {
            double wt = 2.0 * p->wq[i];
            if (wt == 0.0) continue;
            for (int a = 0; a < n; a++)
                for (int b = 0; b <= a; b++) w->Hf[a][b] += wt * w->Sx[k][i][a] * w->Sx[k][i][b];
        }
    }
    for (int k = 0; k < N; k++) {
        if (p->wc[k] != 0.0) {
            double wt = 2.0 * p->wc[k];
            for (int a = 0; a < n; a++) {
                double da = w->Sx[k + 1][5][a] - w->Sx[k][5][a];
                for (int b = 0; b <= a; b++)
                    w->Hf[a][b] += wt * da * (w->Sx[k + 1][5][b] - w->Sx[k][5][b]);
            }
        }
        for (int c = 0; c < 2; c++) w->Hf[iu(w, k) + c][iu(w, k) + c] += 2.0 * p->wr[c];
    }
    /* inequality rows */
    int m = 0;
    for (int k = 0; k < N; k++)
        for (int i = 0; i < 2; i++) {
            w->row[m++] = (rowdef_t){ROW_ULO, k, i, 0};
            w->row[m++] = (rowdef_t){ROW_UHI, k, i, 0};
        }
    for (int k = 1; k <= N; k++) {
        if (p->vlo[k] > -HUGE_VAL) w->row[m++] = (rowdef_t){ROW_XLO, k, 0, 0};
        if (p->vhi[k] < HUGE_VAL) w->row[m++] = (rowdef_t){ROW_XHI, k, 0, 0};
        if (p->elo[k] > -HUGE_VAL) w->row[m++] = (rowdef_t){ROW_XLO, k, 5, 0};
        if (p->ehi[k] < HUGE_VAL) w->row[m++] = (rowdef_t){ROW_XHI, k, 5, 0};
    }
    for (int o2 = 0; o2 < p->nobs; o2++)
        for (int k = 0; k <= N; k++) w->row[m++] = (rowdef_t){ROW_SIG, k, 0, o2};
    for (int o2 = 0; o2 < p->nobs; o2++)
        for (int k = 0; k < N; k++) w->row[m++] = (rowdef_t){ROW_CBF, k, 0, o2};
    w->m = m;
    for (int j = 0; j < m; j++) {
        w->d[j] = 1.0;
        w->J[j][0] = NAN; /* marks "constant row not yet filled" */
    }
}

/* gradient-based scaling of the CBF rows at the starting point, measured in the reference's own
 * (full-space) variables like IPOPT does: d = min(1, gmax / ||grad c||_inf) */
static void scale_rows(work_t* w) {
    const ocp_t* p = w->p;
    for (int j = 0; j < w->m; j++) {
        if (w->row[j].kind != ROW_CBF) continue;
        double dsc, dec, dsn, den;
        int q = p->degree;
        cbf_terms(p, w->x, w->row[j].o, w->row[j].k, &dsc, &dec, &dsn, &den);
        double gm = 1.0; /* |d/d sigma_{i+1}| */
        double v;
        const double Ls = p->Ls[w->row[j].o], Ws = p->Ws[w->row[j].o];
        v = fabs(q * ipow(dsn, q - 1) / Ls); if (v > gm) gm = v;
        v = fabs(q * ipow(den, q - 1) / Ws); if (v > gm) gm = v;
        if (w->row[j].k > 0) { /* x_0 is not a variable */
            v = fabs((1.0 - p->alpha) * q * ipow(dsc, q - 1) / Ls); if (v > gm) gm = v;
            v = fabs((1.0 - p->alpha) * q * ipow(dec, q - 1) / Ws); if (v > gm) gm = v;
        }
        w->d[j] = fmin(1.0, w->o->grad_scale_max / gm);
    }
}

typedef struct {
    int status, iters;
    double kkt, cost;
    double kkt3[3];   /* UNSCALED dual infeasibility / constraint violation / complementarity of the returned iterate (IPOPT's second test) */
} result_t;

/* One workspace per thread, allocated on first use and kept for the life of the thread (OpenMP keeps its
 * pool between parallel regions).  work_t is ~1.3 MB: allocating it inside the parallel region on every
 * call made the page faults of 128 fresh mappings the dominant cost of a 256-problem batch and the
 * all-cores figure SLOWER than one thread. */
static _Thread_local work_t* tl_work = NULL;
static _Thread_local ocp_t* tl_ocp = NULL;
static int thread_ws(work_t** w, ocp_t** p) {
    if (!tl_work) tl_work = (work_t*)malloc(sizeof(work_t));
    if (!tl_ocp) tl_ocp = (ocp_t*)calloc(1, sizeof(ocp_t));
    *w = tl_work; *p = tl_ocp;
    return tl_work && tl_ocp;
}

#include <stdio.h>
static int g_verbose = 0;
void crx_oracle_set_verbose(int v) { g_verbose = v; }
/* diagnostics, mirror of libcrx's crx_debug_kkt_unscaled: kkt[] of a CONVERGED solve becomes 1: the max of / 2: the dual infeasibility /
 * 3: the constraint violation / 4: the complementarity of the returned iterate, UNSCALED (no s_d; CBF rows in the reference's units) */
static int g_kkt_unscaled = 0;
void crx_oracle_debug_kkt_unscaled(int mode) { g_kkt_unscaled = (mode >= 0 && mode <= 4) ? mode : 0; }
/* experiment knobs (tools/tail_knobs.py): 0 JAM_ALPHA, 1 JAM_COUNT, 2 STALL_ITERS, 3 CRAWL_ALPHA, 4 CRAWL_COUNT (0 = off),
 * 5 max restorations, 6 restore at the start when a CBF row of stage <= knob is violated (-1 = off), 7 slack start of a
 * violated row: 0 = |c| (shipped), x > 0 = max(c, x * slack_push) (IPOPT's own start is x = 1), 9 do not charge the knob-6
 * restoration to the budget, 11 probe period of the sticky convexification (0 = CVX_PROBE, < 0 = every iteration probes: not sticky).
 * Defaults = the shipped algorithm; the kernel has no such knobs.  (What used to be knobs 8 and 14 are
 * crx_ipm_opts.reach_screen / .slack_start since ABI 0.2.) */
static double g_knob[16] = {1e-3, 5, 0, 0.0, 0, 2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
void crx_oracle_set_knob(int i, double v) { if (i >= 0 && i < 16) g_knob[i] = v; }

/* Restoration for the CBF NLP, entered when the filter line search finds no acceptable step (where IPOPT switches to
 * its restoration phase).  What jams on crash states (ego inside, or about to enter, an obstacle's unsafe set) is the
 * collapse of the slacks t_j of CBF rows that stay violated.  Every CBF row reads
 *     G_i(x_i, x_{i+1}) + (1 - alpha) sigma_i - sigma_{i+1} >= 0        (control.py:544-558)
 * with free sigma >= 0, so AT THE CURRENT INPUTS the least-violation point IPOPT's restoration looks for
 * (min ||c - t||_1 near the iterate) exists in closed form with zero violation: raise the slacks stage by stage from
 * the end of the horizon, sigma_i >= (sigma_{i+1} - G_i + push_i) / (1 - alpha).  Rows and sigma bounds are then
 * strictly inside, their slacks are re-initialised like at the start (t = c) and their multipliers centred (nu = mu/t);
 * the interior-point iteration resumes from there with a fresh filter.  Returns 0 if nothing changed (no CBF rows,
 * alpha = 1, or every row already feasible): the failure then stands. */
static int restore_slacks(work_t* w, double mu, int full) {
    const ocp_t* p = w->p;
    const crx_ipm_opts* o = w->o;
    const double om = 1.0 - p->alpha;
    if (p->nobs == 0 || !(om > 1e-6)) return 0;
    unpack(w, w->v);
    int changed = 0;
    for (int j = w->m - 1; j >= 0; j--) {     /* CBF rows are stored per obstacle with the stage ascending: walk them backwards */
        if (w->row[j].kind != ROW_CBF) continue;
        const int i = w->row[j].k, ob = w->row[j].o, q = p->degree;
        double dsc, dec, dsn, den;
        cbf_terms(p, w->x, ob, i, &dsc, &dec, &dsn, &den);
        const double G = ipow(dsn, q) + ipow(den, q) - om * (ipow(dsc, q) + ipow(dec, q)) - p->alpha * p->cm;
        const double push = o->slack_push / w->d[j];          /* the SCALED row value ends up >= slack_push */
        const double need = (w->sig[ob][i + 1] - G + push) / om;
        if (need > w->sig[ob][i]) { w->sig[ob][i] = need; w->v[isig(w, i, ob)] = need; changed = 1; }
    }
    if (!changed && !full) return 0;
    eval_full(w);
    for (int j = 0; j < w->m; j++) {
        const int kind = w->row[j].kind;
        if (kind != ROW_CBF && kind != ROW_SIG) {
            /* after a crash restart (full) the inputs have changed as well: every other row restarts with t = max(|c|, push), nu = 1 */
            if (full) { w->t[j] = fmax(fabs(w->c[j]), o->slack_push); w->nu[j] = 1.0; }
            continue;
        }
        w->t[j] = fmax(fabs(w->c[j]), o->slack_push);
        w->nu[j] = fmin(fmax(mu / w->t[j], 1e-8), 1e8);
    }
    return 1;
}

/* Infeasibility certificate for problems whose rows are all linear (planner region QPs, 0-obstacle NLPs) [r2].
 * With y_j >= 0 on the state rows c_j(v) = +-(x_ki(v) - bound) >= 0 the function S(v) = sum_j y_j c_j(v) is linear in the
 * inputs v, and the inputs live in the box [ulo, uhi] (their own rows).  If max over the box of S is negative, no v in
 * the box satisfies every state row: the QP is infeasible (Farkas lemma with the box as the domain) -- a proof, not a
 * heuristic, for ANY y >= 0.  The interior-point multipliers of the violated rows are exactly the y that make S
 * negative: on the BASELINE planner draw the proof exists after 2.7 iterations on average, where the divergence test
 * below (multipliers past 1e12) needs 9.3 and up to 27.  IPOPT reaches the same verdict later ("infeasible problem
 * detected"); the reference only consumes the verdict (overtake_traj_planner.py:359-374: status != success -> the
 * fall-back trajectory).  max_box S = S(v) + sum_a (w_a > 0 ? w_a (uhi_a - v_a) : w_a (ulo_a - v_a)),  w = J_s' y.
 * Returns max_box S; the caller compares it with -1e-8 * max(nu) (rounding in S is ~1e-15 * sum y). */
static double box_certificate(const work_t* w, double* wv) {
    const ocp_t* p = w->p;
    const int n = w->nred, m = w->m;
    double S = 0.0;
    for (int a = 0; a < n; a++) wv[a] = 0.0;
    for (int j = 0; j < m; j++) {
        if (w->row[j].kind != ROW_XLO && w->row[j].kind != ROW_XHI) continue;
        S += w->nu[j] * w->c[j];
        for (int a = 0; a < n; a++) wv[a] += w->J[j][a] * w->nu[j];
    }
    for (int k = 0; k < p->N; k++)
        for (int i = 0; i < 2; i++) {
            const int a = iu(w, k) + i;
            S += wv[a] > 0.0 ? wv[a] * (p->uhi[i] - w->v[a]) : wv[a] * (p->ulo[i] - w->v[a]);
        }
    return S;
}

/* ------------------------------------------------------------------------------------------------
 * Crash path of the MPC-CBF NLP (crx_ipm_opts.slack_start == 2; libcrx: crx_kernels.hip "crash path").
 *
 * Every CBF row reads  G_i(x_i, x_{i+1}) + (1 - alpha) sigma_i - sigma_{i+1} >= 0,  sigma >= 0  (control.py:544-562), so for
 * GIVEN inputs the cheapest slacks are the backward cascade  sigma_N = 0, sigma_i = max(0, (sigma_{i+1} - G_i) / (1 - alpha)),
 * and  phi(u) = f(u) + w sum sigma(u)  is the exact-penalty value of u.  A crash state (ego inside, or about to enter, an
 * obstacle's safety set) needs sigma of 1e2..1e5 (a 1/(1-alpha) growth per stage the car cannot leave the set in); from the
 * reference's start u = 0, sigma = 0 the interior-point iteration crawls there row by row -- each step cut to 1e-2..1e-3 by the
 * fraction-to-the-boundary rule on ONE collapsed slack -- or dies on the way (where IPOPT would enter its restoration phase).
 * Instead such a problem is (re)started from a FEASIBLE INTERIOR point: the best of a G x G grid of constant input pairs
 * (0.9 of the box, states inside their boxes) by phi(u), with its cascade pushed strictly inside.
 * ---------------------------------------------------------------------------------------------- */
#define CRASH_GRID 5
/* G_i of obstacle ob at the rolled-out states w->x */
static double cbf_G(const work_t* w, int ob, int i) {
    const ocp_t* p = w->p;
    const int q = p->degree;
    double dsc, dec, dsn, den;
    cbf_terms(p, w->x, ob, i, &dsc, &dec, &dsn, &den);
    return ipow(dsn, q) + ipow(den, q) - (1.0 - p->alpha) * (ipow(dsc, q) + ipow(dec, q)) - p->alpha * p->cm;
}
/* sum over obstacles and stages of the cascade at w->x; push > 0 keeps every row and every sigma >= push inside; store != 0
 * writes it to w->v */
static double cascade(work_t* w, double push, int store) {
    const ocp_t* p = w->p;
    const double om = 1.0 - p->alpha;
    const int N = p->N;
    double tot = 0.0;
    for (int ob = 0; ob < p->nobs; ob++) {
        double snext = push;
        if (store) w->v[isig(w, N, ob)] = snext;
        tot += snext;
        for (int i = N - 1; i >= 0; i--) {
            double si = (snext - cbf_G(w, ob, i) + push) / om;
            if (si < push) si = push;
            if (store) w->v[isig(w, i, ob)] = si;
            tot += si;
            snext = si;
        }
    }
    return tot;
}
/* reach of s and ey under the boxed inputs: gs[k], ge[k] = sum_{m<k} |e' A^m B| (delta_max, a_max)' */
static void reach_tables(const ocp_t* p, double* gs, double* ge) {
    double ws_[6] = {0, 0, 0, 0, 1, 0}, we_[6] = {0, 0, 0, 0, 0, 1}, as = 0.0, ae = 0.0;
    const double um[2] = {p->uhi[0] > -p->ulo[0] ? p->uhi[0] : -p->ulo[0], p->uhi[1] > -p->ulo[1] ? p->uhi[1] : -p->ulo[1]};
    gs[0] = ge[0] = 0.0;
    for (int k = 1; k <= p->N; k++) {
        double v0 = 0, v1 = 0, e0 = 0, e1 = 0, wn[6] = {0}, en[6] = {0};
        for (int i = 0; i < 6; i++) { v0 += ws_[i] * p->B[i * 2]; v1 += ws_[i] * p->B[i * 2 + 1]; e0 += we_[i] * p->B[i * 2]; e1 += we_[i] * p->B[i * 2 + 1]; }
        as += fabs(v0) * um[0] + fabs(v1) * um[1]; ae += fabs(e0) * um[0] + fabs(e1) * um[1];
        gs[k] = as; ge[k] = ae;
        for (int a = 0; a < 6; a++) for (int i = 0; i < 6; i++) { wn[a] += ws_[i] * p->A[i * 6 + a]; en[a] += we_[i] * p->A[i * 6 + a]; }
        memcpy(ws_, wn, sizeof(wn)); memcpy(we_, en, sizeof(en));
    }
}
/* PROVABLE lower bounds of the slacks at the zero-input roll-out w->x: s_k, ey_k stay within gs[k], ge[k] of the free response, so
 * G_i has an upper bound Gmax_i over ALL admissible inputs and, backwards from L_N = 0, L_i = max(0, (L_{i+1} - Gmax_i) / (1-alpha))
 * bounds sigma_i from below at any feasible point.  Returns whether some L_i > 0; scale != 0 stores scale * L_i as the start
 * (slack_start == 1, libcrx 0.1.3's option). */
static int slack_lower_bounds(work_t* w, double scale) {
    const ocp_t* p = w->p;
    const int N = p->N, q = p->degree;
    const double om = 1.0 - p->alpha;
    double gs[MAXN + 1], ge[MAXN + 1];
    reach_tables(p, gs, ge);
    int any = 0;
    for (int ob = 0; ob < p->nobs; ob++) {
        double Lb = 0.0;
        for (int i = N - 1; i >= 0; i--) {
            double dsc, dec, dsn, den;
            cbf_terms(p, w->x, ob, i, &dsc, &dec, &dsn, &den);
            const double rsc = gs[i] / p->Ls[ob], rec = ge[i] / p->Ws[ob], rsn = gs[i + 1] / p->Ls[ob], ren = ge[i + 1] / p->Ws[ob];
            const double mx_sn = fmax(fabs(dsn - rsn), fabs(dsn + rsn)), mx_en = fmax(fabs(den - ren), fabs(den + ren));
            const double mn_sc = (fabs(dsc) > rsc) ? fabs(dsc) - rsc : 0.0, mn_ec = (fabs(dec) > rec) ? fabs(dec) - rec : 0.0;
            const double Gmax = ipow(mx_sn, q) + ipow(mx_en, q) - om * (ipow(mn_sc, q) + ipow(mn_ec, q)) - p->alpha * p->cm;
            Lb = (Lb - Gmax) / om;
            if (Lb < 0.0) Lb = 0.0;
            if (Lb > 0.0) { any = 1; if (scale != 0.0) w->v[isig(w, i, ob)] = Lb * scale; }
        }
    }
    return any;
}
/* The feasible interior point of the crash path in w->v (inputs + slacks); returns 0 (w->v untouched) when no candidate keeps
 * the states inside their boxes.  Candidate c = a * G + b: delta = 0.9 (2a/(G-1) - 1) delta_max, accel = 0.9 (2b/(G-1) - 1) a_max;
 * value = sum_{k=1..N} sum_i wq_i (x_ki - xr_ki)^2 + sum_{k<N} (wr_0 delta^2 + wr_1 accel^2) + w * cascade (the stage-0 tracking
 * term is the same for all); first minimum wins. */
static int crash_point(work_t* w, int with_cascade) {
    const ocp_t* p = w->p;
    const int N = p->N, n = w->nred, G1 = CRASH_GRID;
    static _Thread_local double vkeep[MAXRED];
    memcpy(vkeep, w->v, sizeof(double) * n);
    double best = HUGE_VAL, bu[2] = {0, 0};
    for (int a = 0; a < G1; a++)
        for (int b = 0; b < G1; b++) {
            const double u0 = 0.9 * (2.0 * a / (G1 - 1) - 1.0) * p->uhi[0], u1 = 0.9 * (2.0 * b / (G1 - 1) - 1.0) * p->uhi[1];
            memset(w->v, 0, sizeof(double) * n);
            for (int k = 0; k < N; k++) { w->v[iu(w, k)] = u0; w->v[iu(w, k) + 1] = u1; }
            unpack(w, w->v);
            int inside = 1;
            double val = 0.0;
            for (int k = 1; k <= N; k++) {
                if (!(w->x[k][0] > p->vlo[k] + 1e-3 && w->x[k][0] < p->vhi[k] - 1e-3 && w->x[k][5] > p->elo[k] + 1e-3 && w->x[k][5] < p->ehi[k] - 1e-3)) inside = 0;
                for (int i = 0; i < 6; i++) { const double e = w->x[k][i] - p->xr[k][i]; val += p->wq[i] * e * e; }
            }
            for (int k = 0; k < N; k++) val += p->wr[0] * u0 * u0 + p->wr[1] * u1 * u1;
            val += p->wsig * cascade(w, 0.0, 0);
            if (inside && val < best) { best = val; bu[0] = u0; bu[1] = u1; }
        }
    if (!(best < HUGE_VAL)) { memcpy(w->v, vkeep, sizeof(double) * n); unpack(w, w->v); return 0; }
    memset(w->v, 0, sizeof(double) * n);
    for (int k = 0; k < N; k++) { w->v[iu(w, k)] = bu[0]; w->v[iu(w, k) + 1] = bu[1]; }
    unpack(w, w->v);
    if (with_cascade) cascade(w, w->o->slack_push, 1);
    else   /* the restart: sigma = push everywhere, restore_slacks() raises the cascade (with its row-scaled push) */
        for (int ob = 0; ob < p->nobs; ob++)
            for (int k = 0; k <= N; k++) w->v[isig(w, k, ob)] = w->o->slack_push;
    unpack(w, w->v);
    return 1;
}
/* slacks and multipliers at the point w->v (start, and the restart of the crash path): t = max(|c|, push) -- inside the bound by at
 * least slack_push and, for a violated row, as large as the violation so that the first fraction-to-the-boundary step is O(1/2),
 * not O(push) --, nu = 1 except on simple-bound rows, which start at the cost gradient that pushes against the bound (dual-feasible
 * start for the 1e4-weighted CBF slacks; IPOPT starts all at 1) */
static void init_rows(work_t* w) {
    const crx_ipm_opts* o = w->o;
    eval_full(w);
    for (int j = 0; j < w->m; j++) {
        w->t[j] = g_knob[7] != 0.0 ? fmax(w->c[j], o->slack_push * g_knob[7]) : fmax(fabs(w->c[j]), o->slack_push);
        w->nu[j] = 1.0;
        const rowdef_t* r = &w->row[j];
        double gg = 0.0;
        if (r->kind == ROW_SIG) gg = w->g[isig(w, r->k, r->o)];
        else if (r->kind == ROW_ULO) gg = w->g[iu(w, r->k) + r->i];
        else if (r->kind == ROW_UHI) gg = -w->g[iu(w, r->k) + r->i];
        if (gg > 1.0) w->nu[j] = gg;
    }
}

static void ipm_solve(work_t* w, result_t* res) {
    const ocp_t* p = w->p;
    const crx_ipm_opts* o = w->o;
    const int n = w->nred, m = w->m;
    const double kappa_sigma = 1e10, smax = 100.0, eta = 1e-8;
    (void)smax;
    memset(w->v, 0, sizeof(double) * n);
    unpack(w, w->v);
    /* crash path: see crash_point() */
    const int crash_path = p->nobs > 0 && o->slack_start >= 2 && o->restore_iters >= 0 && (1.0 - p->alpha) > 1e-6;
    int crash = 0;
    if (p->nobs > 0 && o->slack_start == 1) { slack_lower_bounds(w, 1.0); unpack(w, w->v); }
    /* the kernel searches its candidates BEFORE the loop, for the problems that may take the crash path: a provable crash state (it
     * starts from the point) or a CBF row violated at the zero start (it may stall and restart from the point) */
    int may_restart = 0, crash_at_start = 0;
    if (crash_path) {
        const int crash_state = slack_lower_bounds(w, 0.0);
        for (int ob = 0; ob < p->nobs; ob++)
            for (int i = 0; i < p->N; i++) if (cbf_G(w, ob, i) < 0.0) may_restart = 1;
        /* slack_start == 3 (eager): every problem whose zero start violates a CBF row starts from the point, provable crash state or not --
         * the restart below then never fires.  Faster (headline batch: 30 iterations at most instead of 36, the restart's 14 wasted ones are
         * gone) and further from the reference: a near-miss that the zero start SOLVES ends at the local minimum IPOPT reaches from zero, and at
         * another one from the candidate point (tests/test_gpu_closed_loop.py::test_mpccbf_racing fails with it as the default). */
        if (crash_state || (o->slack_start == 3 && may_restart)) { may_restart = 1; crash = crash_point(w, 1); crash_at_start = crash; }
    }
    scale_rows(w);
    init_rows(w);
    double mu = o->mu_init, dw_last = 0.0, E0 = HUGE_VAL, theta_min = 0.0, theta_max = HUGE_VAL;
    const double CRASH_MU_FRAC = 0.1;
    if (crash_at_start) {
        /* the crash start's barrier parameter comes from its own complementarity: its slacks are 1e2 .. 1e5 with multipliers of 1, and from
         * mu = 0.1 the iteration crawled along the fraction-to-the-boundary rule for 25 steps before mu moved at all (the longest solve of
         * the headline batch: 38 iterations, now 30).  mu_0 = CRASH_MU_FRAC x mean_j(t_j nu_j), not below mu_init; the monotone update
         * takes it down from there (kappa_mu per barrier problem while mu > 1). */
        double sc = 0.0;
        for (int j = 0; j < w->m; j++) sc += w->t[j] * w->nu[j];
        mu = fmin(fmax(CRASH_MU_FRAC * sc / (w->m > 0 ? w->m : 1), o->mu_init), 1e6);
    }
    enum { MAXF = 32 };
    double Fth[MAXF], Fph[MAXF];
    int nf = 0;
    const int JAM_COUNT = (int)g_knob[1], STALL_ITERS = g_knob[2] > 0 ? (int)g_knob[2] : o->stall_iters;   /* knob 2: experiments only (0 = the descriptor's) */
    const double JAM_ALPHA = g_knob[0];
    int status = CRX_MAX_ITER, it = 0, n_restore = 0, first = 1, jam = 0, jam_on = 1, it_limit = 0;
    /* a solve that STARTS on the crash path has a budget too (three times the restoration budget + 1: it has the whole way to go; 2x
     * loses 1 % of the three-car draw), after which it ends CRX_RESTORED like a restarted one -- feasible through its slacks, not
     * optimal -- instead of crawling to max_iter */
    if (crash_at_start) { n_restore = 1; it_limit = 1 + 3 * o->restore_iters; }
    int crawl = 0, cvx_run = 0, short_run = 0;
    double ep_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    enum { CVX_PROBE = 4 };
    static _Thread_local double ctrial[MAXM], ttrial[MAXM], vtrial[MAXRED], rd[MAXRED], rp[MAXM], tmp[MAXRED];
    if (g_knob[6] >= 0.0 && o->restore_iters >= 0) {   /* experiment: slacks first -- restore before the first iteration when a
                                                          CBF row of a stage <= knob is violated at the start point */
        int viol = 0;
        for (int j = 0; j < m; j++)
            if (w->row[j].kind == ROW_CBF && w->row[j].k <= (int)g_knob[6] && w->c[j] < 0.0) viol = 1;
        if (viol && restore_slacks(w, o->mu_init, 0)) { if (g_knob[9] == 0.0) { n_restore = 1; it_limit = 1 + o->restore_iters; } first = 1; }
    }
    for (it = 0;; it++) {
        /* residuals */
        double nus = 0.0;
        for (int j = 0; j < m; j++) nus += fabs(w->nu[j]);
        double sd = fmax(smax, nus / (m > 0 ? m : 1)) / smax;
        double e_d = 0.0, e_p = 0.0, e_c = 0.0;
        for (int a = 0; a < n; a++) {
            double s = w->g[a];
            for (int j = 0; j < m; j++) s -= w->J[j][a] * w->nu[j];
            rd[a] = s;
            (void)rd;
            if (fabs(s) > e_d) e_d = fabs(s);
        }
        e_d /= sd;
        for (int j = 0; j < m; j++) {
            rp[j] = w->c[j] - w->t[j];
            if (fabs(rp[j]) > e_p) e_p = fabs(rp[j]);
            double cc = fabs(w->t[j] * w->nu[j]);
            if (cc > e_c) e_c = cc;
        }
        e_c /= sd;
        E0 = fmax(e_d, fmax(e_p, e_c));
        if (g_verbose)
            fprintf(stderr, "it %3d f %.8e ed %.6e ep %.6e ec %.6e mu %.1e dw %.1e nf %d\n", it, w->f, e_d, e_p, e_c, mu, dw_last, nf);
        /* IPOPT's termination test (IpOptErrorConvCheck.cpp, OptimalityErrorConvergenceCheck::CurrentIsConverged; the reference runs it on
         * default options, control.py:593 / overtake_traj_planner.py:335): the scaled error <= tol AND the UNSCALED dual infeasibility <=
         * dual_inf_tol (1), constraint violation <= constr_viol_tol (1e-4), complementarity <= compl_inf_tol (1e-4).  Unscaled: no s_d, and the
         * gradient-based row scaling d_j of the CBF rows undone (t nu is invariant under it).  The second half binds on crash states:
         * multipliers of 1e7..1e9 make s_d 1e4..1e7 and the scaled complementarity passes at mu = 1e-4 already. */
        res->kkt3[0] = e_d * sd; res->kkt3[2] = e_c * sd; res->kkt3[1] = 0.0;
        for (int j = 0; j < m; j++) res->kkt3[1] = fmax(res->kkt3[1], fabs(rp[j]) / w->d[j]);
        if (E0 <= o->tol && res->kkt3[0] <= o->dual_inf_tol && res->kkt3[1] <= o->constr_viol_tol && res->kkt3[2] <= o->compl_inf_tol) { status = CRX_CONVERGED; break; }
        if (it >= o->max_iter) break;
        if (n_restore > 0 && it >= it_limit) { status = CRX_RESTORED; break; }   /* restoration budget used up */
        /* barrier update */
        for (;;) {
            double e_cm = 0.0;
            for (int j = 0; j < m; j++) {
                double cc = fabs(w->t[j] * w->nu[j] - mu);
                if (cc > e_cm) e_cm = cc;
            }
            e_cm /= sd;
            double Emu = fmax(e_d, fmax(e_p, e_cm));
            if (Emu <= o->kappa_eps * mu && mu > o->tol / 10.0) {
                mu = fmax(o->tol / 10.0, fmin(o->kappa_mu * mu, pow(mu, o->theta_mu)));
                nf = 0; /* new barrier problem: reset the filter */
            } else
                break;
        }
#pragma omp atomic
        g_niter++;
        double tau = fmax(o->tau_min, 1.0 - mu);
        /* H = Hf + curvature + J' Sigma J  (lower triangle) */
        static _Thread_local double Hneg[MAXRED][MAXRED];   /* the negative-semidefinite part of the CBF curvature */
        for (int a = 0; a < n; a++)
            for (int b = 0; b <= a; b++) { w->H[a][b] = w->Hf[a][b]; Hneg[a][b] = 0.0; }
        for (int j = 0; j < m; j++) {
            if (w->row[j].kind != ROW_CBF) continue;
            int i = w->row[j].k, ob = w->row[j].o, q = p->degree;
            double dsc, dec, dsn, den;
            cbf_terms(p, w->x, ob, i, &dsc, &dec, &dsn, &den);
            double wn = w->nu[j] * w->d[j];
            double hsn = q * (q - 1) * ipow(dsn, q - 2) / (p->Ls[ob] * p->Ls[ob]);
            double hen = q * (q - 1) * ipow(den, q - 2) / (p->Ws[ob] * p->Ws[ob]);
            double hsc = q * (q - 1) * ipow(dsc, q - 2) / (p->Ls[ob] * p->Ls[ob]);
            double hec = q * (q - 1) * ipow(dec, q - 2) / (p->Ws[ob] * p->Ws[ob]);
            double om = 1.0 - p->alpha;
            /* W -= nu * hess c */
            for (int a = 0; a < n; a++)
                for (int b = 0; b <= a; b++) {
                    const double ng = wn * (-hsn * w->Sx[i + 1][4][a] * w->Sx[i + 1][4][b] - hen * w->Sx[i + 1][5][a] * w->Sx[i + 1][5][b]);
                    w->H[a][b] += ng + wn * om * (hsc * w->Sx[i][4][a] * w->Sx[i][4][b] + hec * w->Sx[i][5][a] * w->Sx[i][5][b]);
                    Hneg[a][b] += ng;
                }
        }
        for (int j = 0; j < m; j++) {
            double sg = w->nu[j] / w->t[j];
            const double* Jr = w->J[j];
            for (int a = 0; a < n; a++) {
                if (Jr[a] == 0.0) continue;
                double sa = sg * Jr[a];
                for (int b = 0; b <= a; b++) w->H[a][b] += sa * Jr[b];
            }
        }
        for (int a = 0; a < n; a++) {
            double s = -w->g[a];
            for (int j = 0; j < m; j++)
                s += w->J[j][a] * (mu / w->t[j] - w->nu[j] / w->t[j] * rp[j]);
            w->rhs[a] = s;
        }
        /* inertia correction */
        static _Thread_local double Hs[MAXRED][MAXRED];
        for (int a = 0; a < n; a++) memcpy(Hs[a], w->H[a], sizeof(double) * (a + 1));
        double dw = 0.0;
        /* ... and the convexification is STICKY: after an iteration that needed it the next ones start with the convexified matrix, every
         * CVX_PROBE-th of such a run tries the exact one first again (the exact matrix of a crash state fails for a dozen iterations in a
         * row, each time at the far end of the backward sweep: the headline's longest solve spent 17 % of its time on doomed attempts) */
        const int cvxp = g_knob[11] < 0.0 ? 1 : (g_knob[11] > 0.0 ? (int)g_knob[11] : CVX_PROBE);   /* experiments: knob 11 */
        const int start_convex = crash && cvx_run > 0 && (cvx_run % cvxp) != 0;
        int ok = start_convex ? 0 : chol(n, w->H);
        if (ok || !crash) cvx_run = 0;
        if (!ok && crash) {   /* crash path: first retry WITHOUT the reverse-convex part of the CBF curvature (-nu hess g_{i+1}): what
                                 remains is positive definite by construction; IPOPT's delta_w schedule only if that fails */
            cvx_run++;
            for (int a = 0; a < n; a++) {
                memcpy(w->H[a], Hs[a], sizeof(double) * (a + 1));
                for (int b = 0; b <= a; b++) w->H[a][b] -= Hneg[a][b];
            }
            ok = chol(n, w->H);
            dw = -1.0;
            if (!ok)   /* rounding only: IPOPT's schedule goes on from the convexified matrix */
                for (int a = 0; a < n; a++)
                    for (int b = 0; b <= a; b++) Hs[a][b] -= Hneg[a][b];
        }
        if (!ok) {
            dw = dw_last == 0.0 ? 1e-4 : fmax(1e-20, dw_last / 3.0);
            for (;;) {
                for (int a = 0; a < n; a++) {
                    memcpy(w->H[a], Hs[a], sizeof(double) * (a + 1));
                    w->H[a][a] += dw;
                }
                ok = chol(n, w->H);
                if (ok) break;
                dw *= dw_last == 0.0 ? 100.0 : 8.0;
                if (dw > 1e40) break;
            }
            if (!ok) break;
            dw_last = dw;
        }
        if (dw < 0.0) dw = 0.0;
        memcpy(w->dv, w->rhs, sizeof(double) * n);
        chol_solve(n, w->H, w->dv);
        double a_p = 1.0, a_d = 1.0, theta = 0.0, Dphi = 0.0, curv = 0.0;
        int jblock = -1;
        for (int j = 0; j < m; j++) {
            double s = rp[j];
            for (int a = 0; a < n; a++) s += w->J[j][a] * w->dv[a];
            w->dt[j] = s;
            w->dnu[j] = (mu - w->t[j] * w->nu[j] - w->nu[j] * s) / w->t[j];
            if (s < 0.0 && -tau * w->t[j] / s < a_p) { a_p = -tau * w->t[j] / s; jblock = j; }
            if (w->dnu[j] < 0.0) a_d = fmin(a_d, -tau * w->nu[j] / w->dnu[j]);
            theta += fabs(rp[j]);
            Dphi -= mu * s / w->t[j];
        }
        for (int a = 0; a < n; a++) Dphi += w->g[a] * w->dv[a];
        for (int a = 0; a < n; a++) curv += w->dv[a] * w->rhs[a];
        /* filter line search (Waechter & Biegler sec. 2.3; no second-order correction):
         * theta = ||c - t||_1, phi = barrier objective */
        double phi0 = w->f;
        for (int j = 0; j < m; j++) phi0 -= mu * log(w->t[j]);
        if (first) {   /* start, and again after a restoration */
            theta_min = 1e-4 * fmax(1.0, theta);
            theta_max = 1e4 * fmax(1.0, theta);
            first = 0;
        }
        double al = a_p;
        int acc = 0, ftype = 0;
        /* backtracking stops at alpha_min = 1e-10: below it a trial point differs from the iterate by rounding only */
        for (int ls = 0; ls < 40 && al >= 1e-10; ls++) {
            for (int a = 0; a < n; a++) vtrial[a] = w->v[a] + al * w->dv[a];
            double fn;
            eval_fc(w, vtrial, &fn, ctrial);
            double phin = fn, thn = 0.0;
            for (int j = 0; j < m; j++) {
                double tn = w->t[j] + al * w->dt[j];
                if (ctrial[j] > tn) tn = ctrial[j]; /* slack reset: lowers theta and phi */
                ttrial[j] = tn;
                phin -= mu * log(tn);
                thn += fabs(ctrial[j] - tn);
            }
            int okf = (thn <= theta_max) && (phin == phin);
            for (int i = 0; i < nf && okf; i++)
                if (!(thn < Fth[i] || phin < Fph[i])) okf = 0;
            if (okf) {
                int sw = (Dphi < 0.0) && (al * pow(-Dphi, 2.3) > pow(theta, 1.1));
                if (theta <= theta_min && sw) {
                    if (phin <= phi0 + eta * al * Dphi + 10.0 * 2.2e-16 * fabs(phi0)) { acc = 1; ftype = 1; }
                } else if (thn <= (1.0 - 1e-5) * theta || phin <= phi0 - 1e-8 * theta) {
                    acc = 1;
                }
            }
            if (acc) break;
            /* EXPERIMENT (knob 10 = number of corrections, IPOPT: max_soc = 4; 0 = off, the shipped algorithm): IPOPT's second-order correction
             * (Waechter & Biegler 2006, sec. 2.4).  When the FIRST trial step is rejected and does not reduce the constraint violation, the step is
             * re-solved -- same matrix -- with the primal residual replaced by  c_soc = al * (c - t)_k + (c - t)(trial),  up to max_soc times while
             * theta shrinks by kappa_soc = 0.99; a corrected trial point goes through the same filter tests.  Measures what the restatement leaves out. */
            if (ls == 0 && g_knob[10] > 0.0 && thn >= theta) {
                static _Thread_local double csoc[MAXM], dvs[MAXRED], dts[MAXM], vs[MAXRED], cs2[MAXM], ts2[MAXM];
                for (int j = 0; j < m; j++) csoc[j] = al * rp[j] + (ctrial[j] - ttrial[j]);
                double th_old = thn, als = al;
                for (int p_ = 0; p_ < (int)g_knob[10] && !acc; p_++) {
                    for (int a = 0; a < n; a++) {
                        double s_ = -w->g[a];
                        for (int j = 0; j < m; j++) s_ += w->J[j][a] * (mu / w->t[j] - w->nu[j] / w->t[j] * csoc[j]);
                        dvs[a] = s_;
                    }
                    chol_solve(n, w->H, dvs);
                    als = 1.0;
                    for (int j = 0; j < m; j++) {
                        double s_ = csoc[j];
                        for (int a = 0; a < n; a++) s_ += w->J[j][a] * dvs[a];
                        dts[j] = s_;
                        if (s_ < 0.0 && -tau * w->t[j] / s_ < als) als = -tau * w->t[j] / s_;
                    }
                    for (int a = 0; a < n; a++) vs[a] = w->v[a] + als * dvs[a];
                    double fs;
                    eval_fc(w, vs, &fs, cs2);
                    double phs = fs, ths = 0.0;
                    for (int j = 0; j < m; j++) {
                        double tn = w->t[j] + als * dts[j];
                        if (cs2[j] > tn) tn = cs2[j];
                        ts2[j] = tn;
                        phs -= mu * log(tn);
                        ths += fabs(cs2[j] - tn);
                    }
                    int okf2 = (ths <= theta_max) && (phs == phs);
                    for (int i = 0; i < nf && okf2; i++)
                        if (!(ths < Fth[i] || phs < Fph[i])) okf2 = 0;
                    if (okf2) {
                        int sw2 = (Dphi < 0.0) && (al * pow(-Dphi, 2.3) > pow(theta, 1.1));
                        if (theta <= theta_min && sw2) {
                            if (phs <= phi0 + eta * al * Dphi + 10.0 * 2.2e-16 * fabs(phi0)) { acc = 1; ftype = 1; }
                        } else if (ths <= (1.0 - 1e-5) * theta || phs <= phi0 - 1e-8 * theta) {
                            acc = 1;
                        }
                    }
                    if (acc) {
                        memcpy(vtrial, vs, sizeof(double) * n); memcpy(ttrial, ts2, sizeof(double) * m);
                        /* the multiplier step below uses w->dt: the corrected slack step */
                        for (int j = 0; j < m; j++) { w->dt[j] = dts[j]; w->dnu[j] = (mu - w->t[j] * w->nu[j] - w->nu[j] * dts[j]) / w->t[j]; }
                        a_d = 1.0;
                        for (int j = 0; j < m; j++) if (w->dnu[j] < 0.0) a_d = fmin(a_d, -tau * w->nu[j] / w->dnu[j]);
                        al = als;
#pragma omp atomic
                        g_nsoc++;
                        break;
                    }
                    if (ths > 0.99 * th_old) break;
                    th_old = ths;
                    for (int j = 0; j < m; j++) csoc[j] = als * csoc[j] + (cs2[j] - ts2[j]);
                }
                if (acc) break;
            }
            al *= 0.5;
        }
        if (g_verbose > 1 && jblock >= 0)
            fprintf(stderr, "      blocking row %d kind %d k %d i %d: t %.3e dt %.3e c %.3e nu %.3e\n", jblock, w->row[jblock].kind,
                    w->row[jblock].k, w->row[jblock].i, w->t[jblock], w->dt[jblock], w->c[jblock], w->nu[jblock]);
        if (g_verbose) fprintf(stderr, "      a_p %.3e a_d %.6e alpha %.6e acc %d ftype %d theta %.2e Dphi %.2e curv %.2e dw %.1e\n", a_p, a_d, al, acc, ftype, theta, Dphi, curv, dw);
        if (acc && !ftype && nf < MAXF) {
            Fth[nf] = (1.0 - 1e-5) * theta;
            Fph[nf] = phi0 - 1e-8 * theta;
            nf++;
        }
        /* statistic only (crx_oracle_stat(3)): IPOPT's WATCHDOG would arm itself after watchdog_shortened_iter_trigger = 10 successive iterations whose
         * step the line search had to shorten below the fraction-to-the-boundary length -- counted once per solve: an upper bound on the solves the
         * (unrestated) watchdog could influence */
        if (acc && al < a_p) { if (++short_run == 10) {
#pragma omp atomic
            g_nwatch++;
        } } else short_run = 0;
        /* jam: JAM_COUNT accepted steps in a row shorter than JAM_ALPHA while the constraints are still violated -- the
         * slacks of violated CBF rows are collapsing and every step is cut to nothing (IPOPT's alpha < alpha_min test
         * sends it to restoration from the same situation) */
        if (acc && jam_on && al < JAM_ALPHA && e_p > o->tol) jam++; else jam = 0;
        /* crawl: CRAWL_COUNT accepted steps in a row cut below CRAWL_ALPHA by the fraction-to-the-boundary rule while still infeasible */
        if (g_knob[4] > 0) {
            if (acc && jam_on && al < g_knob[3] && e_p > 1e-6) crawl++; else crawl = 0;
            if (crawl >= (int)g_knob[4]) { jam = JAM_COUNT; crawl = 0; }
        }
        /* stall: STALL_ITERS iterations without a restoration and still infeasible -- the same crawl with steps just above JAM_ALPHA.
         * [r5] 100, was 50: calibrated on configs[1] (p99 16 iterations, max 30) the rule stopped HEALTHY three-obstacle N = 20 solves that
         * converge by themselves after 53..86 iterations: 88 of the 135 non-converged problems of the benched configs[3] batch
         * (tests/golden/cfg4_stopped.npz; with the restoration budget at 50 instead of 25: 124 of 135, none lost, configs[1] untouched) */
        if (acc && jam_on && n_restore == 0 && it >= STALL_ITERS && e_p > 1e-6) jam = JAM_COUNT;
        /* theta stagnation (experiment knobs 13 = window W, 14 = ratio R; 0 = off): a zero start that may restart and whose constraint
         * violation fell by less than (1 - R) over the last W iterations */
        {   /* (the window is clamped to the ring's 7 usable slots and the old value is read BEFORE this iteration's is stored: ADVICE r5) */
            const int W = g_knob[13] > 7.0 ? 7 : (int)g_knob[13];
            const double ep_old = ep_hist[(it - W) & 7];
            ep_hist[it & 7] = e_p;
            if (W > 0 && acc && jam_on && n_restore == 0 && crash_path && may_restart && !crash && it >= W && e_p > 1e-6 && e_p > g_knob[14] * ep_old) jam = JAM_COUNT;
        }
        if (!acc || jam >= JAM_COUNT) {
            if (crash_path && may_restart && n_restore < 1 && !crash && crash_point(w, 0)) {
                /* crash path (ii): the solve started at the reference's zero point and stalls on violated CBF rows -- restart ONCE from
                 * the feasible interior point: the candidate's inputs, then the closed-form cascade and the re-initialisation the
                 * restoration uses (all rows) */
                if (g_knob[12] != 0.0) {   /* experiment: the restart as a crash START -- the candidate's cascade, the start's slacks and multipliers, mu from its complementarity */
                    crash_point(w, 1);
                    init_rows(w);
                    double sc = 0.0;
                    for (int j = 0; j < w->m; j++) sc += w->t[j] * w->nu[j];
                    mu = fmin(fmax(CRASH_MU_FRAC * sc / (w->m > 0 ? w->m : 1), o->mu_init), 1e6);
                } else {
                    restore_slacks(w, o->mu_init, 1);
                    mu = o->mu_init;
                }
                if (g_verbose) fprintf(stderr, "      RESTART from the crash point (acc %d jam %d)\n", acc, jam);
                crash = 1;
                n_restore++; it_limit = it + 1 + (g_knob[12] > 1.0 ? 3 : 1) * o->restore_iters;
                nf = 0; first = 1; dw_last = 0.0; jam = 0;
                continue;
            }
            if (o->restore_iters >= 0 && n_restore < (int)g_knob[5] && restore_slacks(w, o->mu_init, 0)) {
                if (g_verbose) fprintf(stderr, "      RESTORE (acc %d jam %d)\n", acc, jam);
                if (n_restore++ == 0) it_limit = it + 1 + o->restore_iters;
                mu = o->mu_init; nf = 0; first = 1; dw_last = 0.0; jam = 0;
                continue;
            }
            if (acc) { jam_on = 0; jam = 0; }   /* nothing to restore: the jam is not about CBF rows; carry on, stop looking */
        }
        if (!acc) {
            /* no acceptable step and nothing to restore: a point of local infeasibility if the constraints are still
             * violated there (IPOPT: "converged to a point of local infeasibility" / "restoration failed") */
            if (e_p > 1e-6) status = CRX_STALLED;   /* IPOPT: "converged to a point of local infeasibility" / "restoration failed": not a proof */
            break;
        }
        memcpy(w->v, vtrial, sizeof(double) * n);
        memcpy(w->t, ttrial, sizeof(double) * m);
        eval_full(w);
        double numax = 0.0, th = 0.0;
        for (int j = 0; j < m; j++) {
            double nn = w->nu[j] + a_d * w->dnu[j];
            nn = fmin(fmax(nn, mu / (kappa_sigma * w->t[j])), kappa_sigma * mu / w->t[j]);
            w->nu[j] = nn;
            if (nn > numax) numax = nn;
            th = fmax(th, fabs(w->c[j] - w->t[j]));
        }
        if (numax > 1e12 && th > 1e-6) { status = CRX_STALLED; it++; break; }   /* IPOPT's divergence heuristic: not a proof */
        /* still violated after the step: look for the proof that it must be (linear rows only; a feasible problem is
         * here 0.14 times per solve on average: the slack reset makes th ~ 0 as soon as a point inside the rows is met) */
        if (p->linear_rows && th > 1e-6 && box_certificate(w, tmp) < -1e-8 * numax) { status = CRX_INFEASIBLE; it++; break; }
    }
    res->status = status;
    res->iters = it;
    res->kkt = E0;
    if (g_kkt_unscaled && status == CRX_CONVERGED)
        res->kkt = g_kkt_unscaled == 1 ? fmax(res->kkt3[0], fmax(res->kkt3[1], res->kkt3[2])) : res->kkt3[g_kkt_unscaled - 2];
    res->cost = w->f;
}

/* ------------------------------------------------------------------------------------------------
 * [r6] Convex rows: Mehrotra's predictor-corrector (Mehrotra, SIAM J. Optim. 2, 1992; the form of Nocedal & Wright, Alg. 16.4) for the
 * problems whose rows are all LINEAR and whose cost is a strictly convex quadratic -- the planner's region QPs
 * (planning/overtake_traj_planner.py:263-334) and MPC-CBF NLPs with no obstacle slot.  Such a problem has ONE solution: parity with the
 * reference does not depend on the path an interior-point method takes to it, only on the tolerance it is solved to (VERDICT r5
 * item 2), so these rows no longer run IPOPT's monotone barrier + filter line search (kept as crx_ipm_opts.qp_method = 1):
 *   per iteration ONE factorisation of  H + J' Sigma J,  Sigma = nu / t,  and two solves with it:
 *   predictor   the affine-scaling step (mu = 0);  step lengths to the boundary;  mu_aff = (t + a_p dt)'(nu + a_d dnu) / m;  sigma = (mu_aff / mu)^3
 *   corrector   the step for  t nu = sigma mu - dt_aff dnu_aff;  fraction-to-the-boundary rule, SEPARATE primal and dual step lengths;
 *               if either comes out below 0.2 the solve is redone (same factor) as a plain centring step  t nu = mu;
 *   no merit function, no filter, no line search: the rows are linear, so the primal residual shrinks by (1 - a_p) and t stays c(z) once a
 *   full primal step was taken.
 * Unchanged around it: the starting point, the error measure and IPOPT's complete termination test (so "converged" means the same thing), the
 * infeasibility proofs (fixed-x0 rows, reachability screen, Farkas certificate over the input box after a step that leaves rows violated), the
 * multiplier-divergence heuristic (CRX_STALLED).  Same arithmetic in crx_kernels.hip (qp_predictor_corrector): the kernel's Riccati recursion
 * is the block Cholesky of the same matrix; it reuses the feedback gains for the second solve.
 * ---------------------------------------------------------------------------------------------- */
#define PC_SHORT_STEP 0.2
static void qp_pc_solve(work_t* w, result_t* res) {
    const crx_ipm_opts* o = w->o;
    const int n = w->nred, m = w->m;
    const double smax = 100.0;
    memset(w->v, 0, sizeof(double) * n);
    unpack(w, w->v);
    scale_rows(w);          /* (no CBF rows here: every d_j stays 1) */
    init_rows(w);
    static _Thread_local double rp[MAXM], dta[MAXM], dna[MAXM], rhs2[MAXRED], tmp[MAXRED];
    int status = CRX_MAX_ITER, it = 0;
    double E0 = HUGE_VAL;
    for (it = 0;; it++) {
        double nus = 0.0;
        for (int j = 0; j < m; j++) nus += fabs(w->nu[j]);
        const double sd = fmax(smax, nus / (m > 0 ? m : 1)) / smax;
        double e_d = 0.0, e_p = 0.0, e_c = 0.0, gap = 0.0;
        for (int a = 0; a < n; a++) {
            double s = w->g[a];
            for (int j = 0; j < m; j++) s -= w->J[j][a] * w->nu[j];
            e_d = fmax(e_d, fabs(s));
        }
        for (int j = 0; j < m; j++) {
            rp[j] = w->c[j] - w->t[j];
            e_p = fmax(e_p, fabs(rp[j]));
            e_c = fmax(e_c, w->t[j] * w->nu[j]);
            gap += w->t[j] * w->nu[j];
        }
        res->kkt3[0] = e_d; res->kkt3[1] = e_p; res->kkt3[2] = e_c;
        E0 = fmax(e_d / sd, fmax(e_p, e_c / sd));
        if (g_verbose) fprintf(stderr, "pc it %3d f %.8e ed %.3e ep %.3e ec %.3e gap/m %.2e\n", it, w->f, e_d / sd, e_p, e_c / sd, gap / (m > 0 ? m : 1));
        if (E0 <= o->tol && e_d <= o->dual_inf_tol && e_p <= o->constr_viol_tol && e_c <= o->compl_inf_tol) { status = CRX_CONVERGED; break; }
        if (it >= o->max_iter) break;
#pragma omp atomic
        g_niter++;
        const double mu = gap / (m > 0 ? m : 1);
        /* H + J' Sigma J (lower triangle), factorised once */
        for (int a = 0; a < n; a++)
            for (int b = 0; b <= a; b++) w->H[a][b] = w->Hf[a][b];
        for (int j = 0; j < m; j++) {
            const double sg = w->nu[j] / w->t[j];
            const double* Jr = w->J[j];
            for (int a = 0; a < n; a++) {
                if (Jr[a] == 0.0) continue;
                const double sa = sg * Jr[a];
                for (int b = 0; b <= a; b++) w->H[a][b] += sa * Jr[b];
            }
        }
        if (!chol(n, w->H)) break;        /* cannot happen for a convex QP short of overflow: the last iterate is returned, CRX_MAX_ITER */
        /* predictor: affine-scaling direction (mu = 0) */
        for (int a = 0; a < n; a++) {
            double s = -w->g[a];
            for (int j = 0; j < m; j++) s += w->J[j][a] * (-w->nu[j] / w->t[j] * rp[j]);
            w->rhs[a] = s;
        }
        memcpy(w->dv, w->rhs, sizeof(double) * n);
        chol_solve(n, w->H, w->dv);
        double ap = 1.0, ad = 1.0;
        for (int j = 0; j < m; j++) {
            double s = rp[j];
            for (int a = 0; a < n; a++) s += w->J[j][a] * w->dv[a];
            dta[j] = s;
            dna[j] = -w->nu[j] - w->nu[j] / w->t[j] * s;
            if (s < 0.0) ap = fmin(ap, -w->t[j] / s);
            if (dna[j] < 0.0) ad = fmin(ad, -w->nu[j] / dna[j]);
        }
        double gap_aff = 0.0;
        for (int j = 0; j < m; j++) gap_aff += (w->t[j] + ap * dta[j]) * (w->nu[j] + ad * dna[j]);
        const double mu_aff = gap_aff / (m > 0 ? m : 1);
        double sigma = mu_aff / mu;
        sigma = sigma * sigma * sigma;
        /* the centering target never goes below IPOPT's smallest barrier parameter (tol / 10): with sigma -> 1e-9 the products t nu fall to 1e-30 in two
         * steps, Sigma = nu / t of the active rows reaches 1e30 and the reduced gradient sits on a rounding floor of 1e-7 -- above tol -- for ever */
        double smu = fmax(sigma * mu, o->tol / 10.0);
        /* corrector: t nu = sigma mu - dt_aff dnu_aff; same factor */
        for (int pass = 0; pass < 2; pass++) {
        for (int a = 0; a < n; a++) {
            double s = w->rhs[a];
            for (int j = 0; j < m; j++) s += w->J[j][a] * ((smu - dta[j] * dna[j]) / w->t[j]);
            rhs2[a] = s;
        }
        memcpy(w->dv, rhs2, sizeof(double) * n);
        chol_solve(n, w->H, w->dv);
        const double tau = fmax(o->tau_min, 1.0 - mu);       /* IPOPT's fraction-to-the-boundary parameter, on the duality measure */
        ap = 1.0; ad = 1.0;
        for (int j = 0; j < m; j++) {
            double s = rp[j];
            for (int a = 0; a < n; a++) s += w->J[j][a] * w->dv[a];
            w->dt[j] = s;
            w->dnu[j] = (smu - dta[j] * dna[j]) / w->t[j] - w->nu[j] - w->nu[j] / w->t[j] * s;
            if (s < 0.0) ap = fmin(ap, -tau * w->t[j] / s);
            if (w->dnu[j] < 0.0) ad = fmin(ad, -tau * w->nu[j] / w->dnu[j]);
        }
        if (pass == 0 && fmin(ap, ad) < PC_SHORT_STEP) {
            /* a short step: the second-order term dt_aff dnu_aff was computed for a full affine step that cannot be taken, and a corrector that leans on
             * it walks the iterate off the central path (1 of 65 536 region QPs of the configs[4] shard cycled for 60 iterations, steps of 0.03..0.1 at
             * sigma 0.5..0.9) -- redo the solve, same factor, as a plain centring step: t nu = mu */
            smu = fmax(mu, o->tol / 10.0);
            for (int j = 0; j < m; j++) dta[j] = 0.0;
            continue;
        }
        break;
        }
        if (g_verbose) fprintf(stderr, "      mu %.2e mu_aff %.2e sigma %.2e a_p %.4f a_d %.4f\n", mu, mu_aff, sigma, ap, ad);
        for (int a = 0; a < n; a++) w->v[a] += ap * w->dv[a];
        eval_full(w);
        double numax = 0.0, th = 0.0;
        for (int j = 0; j < m; j++) {
            double tn = w->t[j] + ap * w->dt[j];
            if (w->c[j] > tn) tn = w->c[j];          /* the slack never lags behind its row (a full primal step makes them equal) */
            w->t[j] = tn;
            w->nu[j] += ad * w->dnu[j];
            numax = fmax(numax, w->nu[j]);
            th = fmax(th, fabs(w->c[j] - tn));
        }
        if (numax > 1e12 && th > 1e-6) { status = CRX_STALLED; it++; break; }   /* IPOPT's divergence heuristic: not a proof */
        if (th > 1e-6 && box_certificate(w, tmp) < -1e-8 * numax) { status = CRX_INFEASIBLE; it++; break; }
    }
    res->status = status;
    res->iters = it;
    res->kkt = E0;
    if (g_kkt_unscaled && status == CRX_CONVERGED)
        res->kkt = g_kkt_unscaled == 1 ? fmax(res->kkt3[0], fmax(res->kkt3[1], res->kkt3[2])) : res->kkt3[g_kkt_unscaled - 2];
    res->cost = w->f;
}

/* ------------------------------------------------------------------------------------------------
 * scipy.interpolate.interp1d(kind="linear") restated (scipy/interpolate/_interpolate.py
 * _call_linear): searchsorted on x, clip index to [1, n-1], slope form.
 * ---------------------------------------------------------------------------------------------- */
static double interp_lin(const double* xs, const double* ys, int n, double x) {
    int hi = 0;
    while (hi < n && xs[hi] < x) hi++; /* searchsorted(xs, x, side="left") */
    if (hi < 1) hi = 1;
    if (hi > n - 1) hi = n - 1;
    int lo = hi - 1;
    double slope = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo]);
    return slope * (x - xs[lo]) + ys[lo];
}

static double clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

void crx_oracle_ipm_opts_default(crx_ipm_opts* o) {
    o->tol = 1e-8; o->max_iter = 200; o->restore_iters = 50; o->mu_init = 0.1; o->kappa_eps = 10.0;
    o->kappa_mu = 0.2; o->theta_mu = 1.5; o->tau_min = 0.99; o->slack_push = 1e-2;
    o->grad_scale_max = 100.0; o->reach_screen = 1; o->slack_start = 2;
    o->dual_inf_tol = 1.0; o->constr_viol_tol = 1e-4; o->compl_inf_tol = 1e-4; o->stall_iters = 100; o->qp_method = 0;
}

/* The region QP of generate_traj_per_region as the canonical stage-structured problem (line numbers into
 * planning/overtake_traj_planner.py). */
static void fill_planner(ocp_t* p, const crx_planner_desc* d, const double* xb, const double* bs, const double* be,
                         const double* ey_lb_b, double ey_ub_b) {
    const int N = d->N;
    memset(p, 0, sizeof(*p));
    p->N = N; p->nobs = 0; p->linear_rows = 1;
    memcpy(p->A, d->A, sizeof(p->A)); memcpy(p->B, d->B, sizeof(p->B));
    memcpy(p->x0, xb, sizeof(p->x0));
    p->wq[4] = d->w_ref; p->wq[5] = d->w_ref;                     /* :333-334 */
    for (int j = 0; j <= N; j++) {
        double st = clip(xb[4] + 1.0 * j * xb[0] * d->dt_ref, bs[0], bs[N]);   /* :330-331 */
        p->xr[j][4] = st;
        p->xr[j][5] = interp_lin(bs, be, N + 1, st);                            /* :332 */
    }
    p->lin[N][4] = -d->w_prog; p->cconst = d->w_prog * xb[4];      /* :328 */
    for (int k = 0; k < N; k++) p->wc[k] = (k >= 1 && k <= N - 2) ? d->w_dey : 0.0; /* :325-327 */
    p->ulo[0] = -d->delta_max; p->uhi[0] = d->delta_max;            /* :280-281 */
    p->ulo[1] = -d->a_max; p->uhi[1] = d->a_max;                    /* :283-284 */
    for (int k = 0; k <= N; k++) {
        p->vlo[k] = -HUGE_VAL; p->vhi[k] = (k >= 1) ? d->vx_max : HUGE_VAL;   /* :276 */
        if (k < N) { p->elo[k] = ey_lb_b[k]; p->ehi[k] = ey_ub_b; } /* :277-324 */
        else { p->elo[k] = -HUGE_VAL; p->ehi[k] = HUGE_VAL; }
    }
}

int crx_oracle_planner_solve(const crx_planner_desc* d, int batch, const double* x0,
                             const double* bez_s, const double* bez_ey, const double* ey_lb,
                             const double* ey_ub, double* X, double* U, double* cost,
                             int32_t* status, double* kkt, int32_t* iters) {
    if (!d || d->N < 2 || d->N > MAXN || batch < 0) return CRX_ERR_ARG;
    const int N = d->N;
#pragma omp parallel
    {
    work_t* w; ocp_t* p;
    const int have_ws = thread_ws(&w, &p);
#pragma omp for schedule(dynamic, 4)
    for (int b = 0; b < batch; b++) {
        if (!have_ws) { status[b] = CRX_MAX_ITER; continue; }
        const double* xb = x0 + 6 * b;
        const double* bs = bez_s + (size_t)(N + 1) * b;
        const double* be = bez_ey + (size_t)(N + 1) * b;
        fill_planner(p, d, xb, bs, be, ey_lb + (size_t)N * b, ey_ub[b]);
        int infeas0 = 0;
        /* rows on the fixed x0 are constants: feasible -> inert, violated -> IPOPT cannot succeed */
        if (xb[5] < p->elo[0] - d->opts.tol || xb[5] > p->ehi[0] + d->opts.tol) infeas0 = 1;
        result_t r;
        /* reachability screen (libcrx: crx_kernels.hip set-up, include/crx.h crx_set_reach_screen): with boxed inputs ey_j stays
         * within reach_j of its free response, reach_j = sum_{m<j} |e_ey' A^m B| (delta_max, a_max)'; a bound on ey_j outside
         * that interval by more than 1e-6 proves the region infeasible before any iteration */
        int screened = 0;
        if (d->opts.reach_screen) {
            double xk[6], wv[6] = {0, 0, 0, 0, 0, 1}, acc = 0.0;
            memcpy(xk, xb, sizeof(xk));
            for (int j = 0; j < N; j++) {
                if (j >= 1) {
                    double v0 = 0.0, v1 = 0.0, wn[6] = {0, 0, 0, 0, 0, 0}, xn[6];
                    for (int i = 0; i < 6; i++) { v0 += wv[i] * d->B[i * 2]; v1 += wv[i] * d->B[i * 2 + 1]; }
                    acc += fabs(v0) * d->delta_max + fabs(v1) * d->a_max;
                    for (int a = 0; a < 6; a++)
                        for (int i = 0; i < 6; i++) wn[a] += wv[i] * d->A[i * 6 + a];
                    memcpy(wv, wn, sizeof(wv));
                    for (int i = 0; i < 6; i++) { double s_ = 0.0; for (int a = 0; a < 6; a++) s_ += d->A[i * 6 + a] * xk[a]; xn[i] = s_; }
                    memcpy(xk, xn, sizeof(xk));
                }
                if (p->elo[j] > xk[5] + acc + 1e-6 || p->ehi[j] < xk[5] - acc - 1e-6) screened = 1;
            }
        }
        if (screened) { r.status = CRX_INFEASIBLE; r.iters = 0; r.kkt = INFINITY; r.cost = INFINITY; }
        else {
        setup(w, p, &d->opts);
        if (d->opts.qp_method == 0) qp_pc_solve(w, &r); else ipm_solve(w, &r);
        }
        if (infeas0) r.status = CRX_INFEASIBLE;
        double* Xb = X + (size_t)(N + 1) * 6 * b;
        double* Ub = U + (size_t)N * 2 * b;
        if (r.status == CRX_CONVERGED) {
            unpack(w, w->v);
            memcpy(Xb, w->x, sizeof(double) * 6 * (N + 1));
            memcpy(Ub, w->u, sizeof(double) * 2 * N);
            cost[b] = r.cost;
        } else { /* :365-374 */
            memset(Xb, 0, sizeof(double) * 6 * (N + 1));
            memset(Ub, 0, sizeof(double) * 2 * N);
            for (int j = 0; j <= N; j++) {
                /* xcurv_ego (wrapped, :366) differs from ego.xcurv only in s; the wrapped s is the first Bezier
                 * control point (planner_helper.py:49) */
                double st = bs[0] + d->fallback_gain * j * d->dt_ref * xb[0];
                Xb[6 * j + 0] = d->fallback_gain * xb[0];
                Xb[6 * j + 4] = st;
                Xb[6 * j + 5] = interp_lin(bs, be, N + 1, clip(st, bs[0], bs[N]));
            }
            cost[b] = HUGE_VAL;
        }
        status[b] = r.status; kkt[b] = r.kkt; iters[b] = r.iters;
    }
    }
    return CRX_OK;
}

/* control.mpccbf / control.mpc_multi_agents as the canonical stage-structured problem (line numbers into
 * control/control.py).  xt_b: [6] or, with per_stage_target, [N+1][6]; obstacle arrays: this problem's [V][N+1] block. */
static void fill_cbf(ocp_t* p, const crx_cbf_desc* d, const double* xb, const double* xt_b, const double* obs_s_b,
                     const double* obs_ey_b, const double* lap_off_b, int nobs, const double* dims_b) {
    const int N = d->N, V = d->n_obs_max;
    memset(p, 0, sizeof(*p));
    p->N = N; p->nobs = nobs; p->linear_rows = (V == 0);
    memcpy(p->A, d->A, sizeof(p->A)); memcpy(p->B, d->B, sizeof(p->B));
    memcpy(p->x0, xb, sizeof(p->x0));
    memcpy(p->wq, d->Q, sizeof(p->wq));                            /* :588-591 */
    for (int k = 0; k <= N; k++)
        memcpy(p->xr[k], d->per_stage_target ? xt_b + (size_t)k * 6 : xt_b,
               sizeof(double) * 6);
    p->wr[0] = d->R[0]; p->wr[1] = d->R[1];                        /* :578-579 */
    p->wsig = d->w_slack;                                          /* :560,562 */
    p->ulo[0] = -d->delta_max; p->uhi[0] = d->delta_max;           /* :572-573 */
    p->ulo[1] = -d->a_max; p->uhi[1] = d->a_max;                   /* :575-576 */
    for (int k = 0; k <= N; k++) {                                 /* :582-586 */
        p->vlo[k] = d->v_min; p->vhi[k] = d->v_max; p->elo[k] = -d->ey_max; p->ehi[k] = d->ey_max;
    }
    for (int o = 0; o < p->nobs; o++) {
        memcpy(p->obs_s[o], obs_s_b + (size_t)o * (N + 1), sizeof(double) * (N + 1));
        memcpy(p->obs_ey[o], obs_ey_b + (size_t)o * (N + 1), sizeof(double) * (N + 1));
        p->lap_off[o] = lap_off_b[o];
    }
    p->alpha = d->alpha; p->cm = 1.0 + d->margin;
    for (int o = 0; o < MAXO; o++) {   /* :529-535: per obstacle when the caller gives them */
        p->Ls[o] = (dims_b && o < nobs) ? dims_b[2 * o] : d->l_sum;
        p->Ws[o] = (dims_b && o < nobs) ? dims_b[2 * o + 1] : d->w_sum;
    }
    p->degree = d->degree;
}

int crx_oracle_cbf_solve_dims(const crx_cbf_desc* d, int batch, const double* x0, const double* xt,
                              const double* obs_s, const double* obs_ey, const double* lap_off,
                              const int32_t* n_obs, const double* obs_dims, double* X, double* U, double* sigma, double* cost,
                              int32_t* status, double* kkt, int32_t* iters);
int crx_oracle_cbf_solve(const crx_cbf_desc* d, int batch, const double* x0, const double* xt,
                         const double* obs_s, const double* obs_ey, const double* lap_off,
                         const int32_t* n_obs, double* X, double* U, double* sigma, double* cost,
                         int32_t* status, double* kkt, int32_t* iters) {
    return crx_oracle_cbf_solve_dims(d, batch, x0, xt, obs_s, obs_ey, lap_off, n_obs, NULL, X, U, sigma, cost, status, kkt, iters);
}

int crx_oracle_cbf_solve_dims(const crx_cbf_desc* d, int batch, const double* x0, const double* xt,
                              const double* obs_s, const double* obs_ey, const double* lap_off,
                              const int32_t* n_obs, const double* obs_dims, double* X, double* U, double* sigma, double* cost,
                              int32_t* status, double* kkt, int32_t* iters) {
    if (!d || d->N < 2 || d->N > MAXN || batch < 0 || d->n_obs_max < 0 || d->n_obs_max > MAXO ||
        (d->degree & 1) || d->degree < 2)
        return CRX_ERR_ARG;
    const int N = d->N, V = d->n_obs_max;
    if (n_obs)
        for (int b = 0; b < batch; b++)
            if (n_obs[b] < 0 || n_obs[b] > V) return CRX_ERR_ARG;
#pragma omp parallel
    {
    work_t* w; ocp_t* p;
    const int have_ws = thread_ws(&w, &p);
#pragma omp for schedule(dynamic, 4)
    for (int b = 0; b < batch; b++) {
        if (!have_ws) { status[b] = CRX_MAX_ITER; continue; }
        const double* xb = x0 + 6 * b;
        fill_cbf(p, d, xb, d->per_stage_target ? xt + (size_t)(N + 1) * 6 * b : xt + 6 * b, obs_s + (size_t)V * (N + 1) * b,
                 obs_ey + (size_t)V * (N + 1) * b, lap_off + (size_t)V * b, n_obs ? n_obs[b] : V, obs_dims ? obs_dims + (size_t)2 * V * b : NULL);
        int infeas0 = (xb[0] < d->v_min - d->opts.tol || xb[0] > d->v_max + d->opts.tol ||
                       xb[5] < -d->ey_max - d->opts.tol || xb[5] > d->ey_max + d->opts.tol); /* Q9 */
        result_t r;
        setup(w, p, &d->opts);
        if (p->linear_rows && d->opts.qp_method == 0) qp_pc_solve(w, &r); else ipm_solve(w, &r);
        if (infeas0) r.status = CRX_INFEASIBLE;
        unpack(w, w->v);
        memcpy(X + (size_t)(N + 1) * 6 * b, w->x, sizeof(double) * 6 * (N + 1));
        memcpy(U + (size_t)N * 2 * b, w->u, sizeof(double) * 2 * N);
        for (int o = 0; o < V; o++)
            for (int k = 0; k <= N; k++)
                sigma[((size_t)V * b + o) * (N + 1) + k] = o < p->nobs ? w->sig[o][k] : 0.0;
        cost[b] = r.cost; status[b] = r.status; kkt[b] = r.kkt; iters[b] = r.iters;
    }
    }
    return CRX_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Problem-row probes (tests only): evaluate the problem this file builds from C-ABI inputs at a caller-chosen point
 * (inputs U, slacks sigma; states by the roll-out), so that tests can compare cost and rows with what the
 * reference's own code built for the same scenario (tests/golden/cfg*_draw.npz hold the reference's values at the
 * same point), independently of any solve.
 *   cost      the objective
 *   cbf_rows  [n_obs_max][N]   UNSCALED values of  h_{i+1} - h_i + alpha h_i  (control.py:559), 0 for absent obstacles
 *   box       [4][N+1][... ] -> flat: ulo[2], uhi[2], then vlo[N+1], vhi[N+1], elo[N+1], ehi[N+1] (+-inf = absent)
 *   X         [N+1][6] the rolled-out states
 * ---------------------------------------------------------------------------------------------- */
static void probe_common(work_t* w, ocp_t* p, const crx_ipm_opts* o, const double* U, const double* sigma, int V,
                         double* cost, double* cbf_rows, double* box, double* X) {
    const int N = p->N;
    setup(w, p, o);
    for (int k = 0; k < N; k++) { w->v[iu(w, k)] = U[2 * k]; w->v[iu(w, k) + 1] = U[2 * k + 1]; }
    for (int ob = 0; ob < p->nobs; ob++)
        for (int k = 0; k <= N; k++) w->v[isig(w, k, ob)] = sigma[(size_t)ob * (N + 1) + k];
    unpack(w, w->v);
    *cost = cost_value(w);
    if (cbf_rows) {
        for (int i = 0; i < V * N; i++) cbf_rows[i] = 0.0;
        for (int j = 0; j < w->m; j++)
            if (w->row[j].kind == ROW_CBF) cbf_rows[(size_t)w->row[j].o * N + w->row[j].k] = row_value(w, j);
    }
    double* b = box;
    for (int i = 0; i < 2; i++) *b++ = p->ulo[i];
    for (int i = 0; i < 2; i++) *b++ = p->uhi[i];
    for (int k = 0; k <= N; k++) *b++ = p->vlo[k];
    for (int k = 0; k <= N; k++) *b++ = p->vhi[k];
    for (int k = 0; k <= N; k++) *b++ = p->elo[k];
    for (int k = 0; k <= N; k++) *b++ = p->ehi[k];
    memcpy(X, w->x, sizeof(double) * 6 * (N + 1));
}

int crx_oracle_cbf_probe(const crx_cbf_desc* d, const double* x0, const double* xt, const double* obs_s,
                         const double* obs_ey, const double* lap_off, int n_obs, const double* obs_dims, const double* U,
                         const double* sigma, double* cost, double* cbf_rows, double* box, double* X) {
    if (!d || d->N < 2 || d->N > MAXN || d->n_obs_max < 0 || d->n_obs_max > MAXO || n_obs < 0 || n_obs > d->n_obs_max)
        return CRX_ERR_ARG;
    work_t* w; ocp_t* p;
    if (!thread_ws(&w, &p)) return CRX_ERR_ARG;
    fill_cbf(p, d, x0, xt, obs_s, obs_ey, lap_off, n_obs, obs_dims);
    probe_common(w, p, &d->opts, U, sigma, d->n_obs_max, cost, cbf_rows, box, X);
    return CRX_OK;
}

int crx_oracle_planner_probe(const crx_planner_desc* d, const double* x0, const double* bez_s, const double* bez_ey,
                             const double* ey_lb, double ey_ub, const double* U, double* cost, double* box, double* X) {
    if (!d || d->N < 2 || d->N > MAXN) return CRX_ERR_ARG;
    work_t* w; ocp_t* p;
    if (!thread_ws(&w, &p)) return CRX_ERR_ARG;
    fill_planner(p, d, x0, bez_s, bez_ey, ey_lb, ey_ub);
    probe_common(w, p, &d->opts, U, NULL, 0, cost, NULL, box, X);
    return CRX_OK;
}

#ifdef _OPENMP
#include <omp.h>
int crx_oracle_threads(void) { return omp_get_max_threads(); }
void crx_oracle_set_threads(int n) { omp_set_num_threads(n); }
#else
int crx_oracle_threads(void) { return 1; }
void crx_oracle_set_threads(int n) { (void)n; }
#endif

/* The front of OvertakeTrajPlanner: planning/overtake_traj_planner.py:29-42 (get_overtake_flag), :66-76 (partial ey sort,
 * quirk Q3), :77-92 (predictions, veh_infos in iteration order, quirk Q4); planning/planner_helper.py:218-266
 * (check_ego_agent_distance), :177-201 (get_agent_info: max_delta_v). */
int crx_oracle_planner_scene(const crx_scene_desc* d, int n_scen, const double* ego_xcurv, const int32_t* n_all,
                             const double* veh_xcurv, const double* pred_s, const double* pred_ey, int32_t* n_veh,
                             int32_t* overflow, int32_t* order, double* veh_info, double* max_dv, double* obs_s, double* obs_ey) {
    if (!d || d->N < 1 || d->N > MAXN || d->n_veh_max < 1 || d->n_veh_max > MAXO || d->n_all_max < 1 || n_scen < 0) return CRX_ERR_ARG;
    const int N1 = d->N + 1, VA = d->n_all_max, V = d->n_veh_max;
    const double L = d->lap_length;
    for (int s = 0; s < n_scen; s++) {
        const double* ego = ego_xcurv + (size_t)6 * s;
        const double* vx = veh_xcurv + (size_t)6 * VA * s;
        const int na = n_all[s];
        if (na < 0 || na > VA) return CRX_ERR_ARG;
        double s_e = ego[4];
        while (s_e > L) s_e -= L;                                                    /* planner_helper.py:222-223 */
        int idx[MAXO], nv = 0, over = 0, nh = 0;
        char hit_[64] = {0}, keep_[64] = {0};
        double gap_[64];
        if (VA > 64) return CRX_ERR_ARG;
        for (int v = 0; v < na; v++) {
            const double dv = fabs(ego[0] - vx[6 * v]);
            double s_a = vx[6 * v + 4];
            while (s_a > L) s_a -= L;
            const double ahead = d->safety_factor * d->veh_length + d->prediction_factor * dv, behind = 1.0 * d->veh_length;
            const int hit = (s_a - s_e <= ahead && s_a >= s_e) || (s_a + L - s_e <= ahead && s_a + L >= s_e) ||
                            (s_e - s_a <= behind && s_a <= s_e) || (s_e + L - s_a <= behind && s_a <= s_e + L);
            const double g = s_a - s_e;
            gap_[v] = fmin(fabs(g), fmin(fabs(g + L), fabs(g - L)));
            if (hit) { hit_[v] = 1; nh++; }
        }
        if (nh > V) {   /* more vehicles of interest than slots (the reference has no limit): the V nearest, in dict order */
            for (int k = 0; k < V; k++) {
                int best = -1;
                for (int v = 0; v < na; v++)
                    if (hit_[v] && !keep_[v] && (best < 0 || gap_[v] < gap_[best])) best = v;
                keep_[best] = 1;
            }
            over = nh - V;
            memcpy(hit_, keep_, sizeof(hit_));
        }
        for (int v = 0; v < na; v++)
            if (hit_[v]) idx[nv++] = v;
        int ord[MAXO];
        for (int k = 0; k < nv; k++) {                                                /* overtake_traj_planner.py:70-76 */
            const double e = vx[6 * idx[k] + 5];
            if (k == 0) ord[0] = idx[0];
            else if (e >= vx[6 * ord[0] + 5]) { for (int q = k; q > 0; q--) ord[q] = ord[q - 1]; ord[0] = idx[k]; }
            else ord[k] = idx[k];
        }
        double mdv = 0.0;
        for (int k = 0; k < nv; k++) mdv = fmax(mdv, fabs(ego[0] - vx[6 * ord[k]]));
        n_veh[s] = nv; overflow[s] = over; max_dv[s] = mdv;
        for (int k = 0; k < V; k++) {
            order[(size_t)V * s + k] = k < nv ? ord[k] : -1;
            double* vi = veh_info + ((size_t)s * V + k) * 3;
            vi[0] = vi[1] = vi[2] = 0.0;
            if (k < nv) {                                                             /* :87-92, iteration order */
                const double* pe = pred_ey + ((size_t)s * VA + idx[k]) * N1;
                vi[0] = vx[6 * idx[k] + 4]; vi[1] = pe[0]; vi[2] = pe[0];
                for (int j = 1; j < N1; j++) { vi[1] = fmax(vi[1], pe[j]); vi[2] = fmin(vi[2], pe[j]); }
            }
            for (int j = 0; j < N1; j++) {
                obs_s[((size_t)s * V + k) * N1 + j] = k < nv ? pred_s[((size_t)s * VA + ord[k]) * N1 + j] : 0.0;
                obs_ey[((size_t)s * V + k) * N1 + j] = k < nv ? pred_ey[((size_t)s * VA + ord[k]) * N1 + j] : 0.0;
            }
        }
    }
    return CRX_OK;
}

/* planning/overtake_traj_planner.py:205-246 */
int crx_oracle_select(const crx_select_desc* d, int n_scen, const int32_t* n_veh, const double* X,
                      const double* obs_s, const double* obs_ey, const int32_t* old_flag,
                      int32_t* flag, double* sel_cost, double* best_X) {
    if (!d || d->N < 1 || d->N > MAXN || d->n_veh_max < 0 || d->n_veh_max > MAXO) return CRX_ERR_ARG;
    const int N = d->N, V = d->n_veh_max, R = V + 1;
    const double r2 = d->veh_length * d->veh_length + d->veh_width * d->veh_width;
    for (int s = 0; s < n_scen; s++) {
        int nv = n_veh[s];
        if (nv < 0 || nv > V) return CRX_ERR_ARG;
        int best = 0;
        double bestc = HUGE_VAL;
        for (int r = 0; r < R; r++) {
            double c = HUGE_VAL;
            if (r <= nv) {
                const double* Xr = X + (((size_t)s * R + r) * (N + 1)) * 6;
                c = -d->w_prog * (Xr[6 * N + 4] - Xr[4]);                                /* :209 */
                for (int side = 0; side < 2; side++) {
                    int v = side == 0 ? r - 1 : r;                                       /* :213, :227 */
                    if (v < 0 || v >= nv) continue;
                    for (int j = 0; j <= N; j++) {
                        double os = obs_s[((size_t)s * V + v) * (N + 1) + j];
                        while (os > d->lap_length) os -= d->lap_length;                  /* :216-217 */
                        double ds = Xr[6 * j + 4] - os;
                        double de = Xr[6 * j + 5] - obs_ey[((size_t)s * V + v) * (N + 1) + j];
                        if (!(ds * ds + de * de - r2 >= 0.0)) c += d->w_coll;            /* :220-223 */
                    }
                }
                if (old_flag[s] >= 0 && old_flag[s] != r) c += d->w_switch;              /* :238-243 */
                if (c < bestc) { bestc = c; best = r; }                                  /* first arg-min :244 */
            }
            sel_cost[(size_t)s * R + r] = c;
        }
        flag[s] = best;
        memcpy(best_X + (size_t)s * (N + 1) * 6, X + (((size_t)s * R + best) * (N + 1)) * 6,
               sizeof(double) * 6 * (N + 1));
    }
    return CRX_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Plant: DynamicBicycleModel.forward_dynamics with zero noise
 * (utils/base.py:897-942) = n_sub explicit Euler sub-steps of system/vehicle_dynamics.py:4-49 with the
 * curvature looked up per sub-step (utils/racing_env.py:225-246: s wrapped into one lap, first segment
 * with lo <= s <= hi).  Pinned by tests/golden/harness.npz (single steps and a PID closed loop recorded
 * from the reference).
 * ---------------------------------------------------------------------------------------------- */
int crx_oracle_plant_step_noise(const crx_plant_desc* d, int batch, const double* track, const double* xglob, const double* xcurv,
                                const double* u, const double* noise_z, double* xglob_next, double* xcurv_next);
int crx_oracle_plant_step(const crx_plant_desc* d, int batch, const double* track, const double* xglob,
                          const double* xcurv, const double* u, double* xglob_next, double* xcurv_next) {
    return crx_oracle_plant_step_noise(d, batch, track, xglob, xcurv, u, NULL, xglob_next, xcurv_next);
}

/* ... with the bounded process noise of utils/base.py:929-939 from caller-supplied standard-normal draws noise_z [batch][3]
 * (np.random.randn() in the reference, in the order vx, vy, wz); NULL = zero noise */
int crx_oracle_plant_step_noise(const crx_plant_desc* d, int batch, const double* track, const double* xglob, const double* xcurv,
                                const double* u, const double* noise_z, double* xglob_next, double* xcurv_next) {
    if (!d || batch < 0 || d->n_seg < 1) return CRX_ERR_ARG;
    for (int b = 0; b < batch; b++) {
        double vx = xcurv[6 * b], vy = xcurv[6 * b + 1], wz = xcurv[6 * b + 2];
        double epsi = xcurv[6 * b + 3], s = xcurv[6 * b + 4], ey = xcurv[6 * b + 5];
        double psi = xglob[6 * b + 3], X = xglob[6 * b + 4], Y = xglob[6 * b + 5];
        const double delta = u[2 * b], acc = u[2 * b + 1], dt = d->dt_sub;
        for (int it = 0; it < d->n_sub; it++) {
            double sw = s, curv = 0.0;
            while (sw > d->lap_length) sw -= d->lap_length;
            while (sw < 0.0) sw += d->lap_length;
            for (int i = 0; i < d->n_seg; i++)
                if (sw >= track[6 * i + 3] && sw <= track[6 * i + 3] + track[6 * i + 4]) { curv = track[6 * i + 5]; break; }
            const double slip_f = delta - atan2(vy + d->lf * wz, vx);          /* :24 */
            const double slip_r = -atan2(vy - d->lf * wz, vx);                 /* :25 (lf, as the reference) */
            const double Fyf = 2 * d->Df * sin(d->Cf * atan(d->Bf * slip_f));  /* :27 */
            const double Fyr = 2 * d->Dr * sin(d->Cr * atan(d->Br * slip_r));  /* :28 */
            const double dvx = acc - 1 / d->m * Fyf * sin(delta) + wz * vy;    /* :31 */
            const double dvy = 1 / d->m * (Fyf * cos(delta) + Fyr) - wz * vx;  /* :32 */
            const double dwz = 1 / d->Iz * (d->lf * Fyf * cos(delta) - d->lr * Fyr); /* :33 */
            const double v_long = (vx * cos(epsi) - vy * sin(epsi)) / (1 - curv * ey);
            const double n_psi = psi + dt * wz;
            const double n_X = X + dt * (vx * cos(psi) - vy * sin(psi)), n_Y = Y + dt * (vx * sin(psi) + vy * cos(psi));
            const double n_epsi = epsi + dt * (wz - v_long * curv), n_s = s + dt * v_long;
            const double n_ey = ey + dt * (vx * sin(epsi) + vy * cos(epsi));
            const double n_vx = vx + dt * dvx, n_vy = vy + dt * dvy, n_wz = wz + dt * dwz;
            vx = n_vx; vy = n_vy; wz = n_wz; epsi = n_epsi; s = n_s; ey = n_ey; psi = n_psi; X = n_X; Y = n_Y;
        }
        double* g = xglob_next + 6 * (size_t)b;
        double* c = xcurv_next + 6 * (size_t)b;
        g[0] = vx; g[1] = vy; g[2] = wz; g[3] = psi; g[4] = X; g[5] = Y;
        if (noise_z) {   /* :929-939: clipped, HALF added, curvilinear copy only */
            const double* z = noise_z + 3 * (size_t)b;
            vx += 0.5 * fmax(-0.05, fmin(z[0] * 0.01, 0.05));
            vy += 0.5 * fmax(-0.1, fmin(z[1] * 0.01, 0.1));
            wz += 0.5 * fmax(-0.05, fmin(z[2] * 0.005, 0.05));
        }
        c[0] = vx; c[1] = vy; c[2] = wz; c[3] = epsi; c[4] = s; c[5] = ey;
    }
    return CRX_OK;
}
