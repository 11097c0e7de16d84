// crx_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of libcrx.
//
// One finite-horizon optimal-control problem per 64-lane wavefront (one single-wave workgroup per
// problem).  The whole interior-point solve -- iterate, slacks, multipliers, filter, Riccati
// factors -- lives in that workgroup's LDS slice; HBM is touched twice: one coalesced read of the
// problem's inputs, one coalesced write of its trajectory.  FP64 throughout (the reduced Hessians
// have condition numbers 1e6..1e8 and the degree-6 barrier rows reach 1e10).  No MFMA:
// the largest dense block is 9x14.
//
// Algorithm (the same mathematical iteration as oracle/crx_oracle.c, which factorises
// the condensed Newton system with a dense Cholesky instead):
//   primal-dual interior point on  min f(z)  s.t. c_j(z) - t_j = 0, t_j >= 0  for every inequality,
//   dynamics kept satisfied exactly (linear, x0 fixed), Newton step by a Riccati recursion over the
//   augmented stage state (x_k, sigma_k) / input (u_k, sigma_{k+1}), inertia correction through the
//   Riccati pivots, monotone barrier update, fraction-to-the-boundary rule, filter line search.
//
// What each block restates (paths into /root/reference/car_racing):
//   planner region QP   planning/overtake_traj_planner.py:263-334, fall-back :365-374
//   MPC-CBF NLP         control/control.py:492-591 (mpccbf), :270-382 (mpc_multi_agents)
//   region selection    planning/overtake_traj_planner.py:205-246
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "crx_kparams.h"

#define WAVE 64
#define MAXF 32 /* filter entries */

#define SYNC() __syncthreads()

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}

// sum over the wave of log(v) for positive v (lanes with nothing to add pass 1.0): mantissas are
// multiplied (a product of up to 6*64 values in [0.5,1) cannot underflow a double: 2^-384),
// exponents are added, and ONE log is taken.
struct LogAcc {
    double m;
    int e;
    __device__ __forceinline__ LogAcc() : m(1.0), e(0) {}
    __device__ __forceinline__ void mul(double v) {
        int ex;
        double mm = frexp(v, &ex);
        m *= mm;
        e += ex;
    }
    __device__ __forceinline__ double wave_total() {
        double mm = m;
        int ee = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mm *= __shfl_xor(mm, o);
            ee += __shfl_xor(ee, o);
        }
        return log(mm) + 0.6931471805599453 * (double)ee;
    }
};

__device__ __forceinline__ double ipow_d(double a, int p) {
    double r = 1.0;
    for (int i = 0; i < p; i++) r *= a;
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

template <int NOBS>
struct Dim {
    static constexpr int NX = 6 + NOBS;          // augmented state (x, sigma_k)
    static constexpr int NU = 2 + NOBS;          // augmented input (u, sigma_{k+1})
    static constexpr int NZ = NX + NU;
    static constexpr int NR = 8 + 2 * NOBS;      // inequality rows owned by a stage
    // row slots inside a stage: 0..3 input box (d lo, d hi, a lo, a hi); 4..7 box of x_{k+1}
    // (vx lo, vx hi, ey lo, ey hi); 8+o: sigma_{k+1}^o >= 0; 8+NOBS+o: CBF row (k, o)
};

// LDS carve-up (doubles).  Everything is sized from the run-time horizon N.
template <int NOBS>
struct Lds {
    using D = Dim<NOBS>;
    double *M, *x, *u, *sg, *dx, *du, *xr, *vlo, *vhi, *elo, *ehi, *wc, *obs_s, *obs_e;
    double *rt, *rnu, *rdt, *rdnu, *rc, *rd, *rtt;
    double *Hd, *hg, *ga, *Jc, *wJ, *kS, *kE, *kC, *gC;
    double *P, *pv, *T, *H, *hv, *Kk, *kf, *lam, *Fth, *Fph, *cst;
    int m;
    __host__ __device__ static int doubles(int N) {
        int m = N * D::NR + NOBS;
        int n = D::NX * D::NZ;                       // M
        n += (N + 1) * 6 + N * 2 + (N + 1) * (NOBS ? NOBS : 1);  // x u sg
        n += (N + 1) * D::NX + N * D::NU;            // dx du
        n += (N + 1) * 6 + 4 * (N + 1) + N;          // xr, bounds, wc
        n += 2 * (NOBS ? NOBS : 1) * (N + 1);        // obstacles
        n += 7 * m;                                  // rows
        n += 3 * (N + 1) * D::NZ;                    // Hd hg ga
        n += N * (NOBS ? NOBS : 1) * D::NZ + N * (NOBS ? NOBS : 1) + 4 * N;  // Jc wJ kS kE kC gC
        n += D::NX * D::NX + D::NX + D::NX * D::NZ + D::NZ * D::NZ + D::NZ;  // P pv T H hv
        n += N * D::NU * D::NX + N * D::NU + D::NZ;  // Kk kf lam
        n += 2 * MAXF + 16;
        return n;
    }
    __device__ void carve(double* base, int N) {
        const int no = NOBS ? NOBS : 1;
        m = N * D::NR + NOBS;
        double* q = base;
        auto take = [&](int n) { double* r = q; q += n; return r; };
        M = take(D::NX * D::NZ);
        x = take((N + 1) * 6); u = take(N * 2); sg = take((N + 1) * no);
        dx = take((N + 1) * D::NX); du = take(N * D::NU);
        xr = take((N + 1) * 6);
        vlo = take(N + 1); vhi = take(N + 1); elo = take(N + 1); ehi = take(N + 1); wc = take(N);
        obs_s = take(no * (N + 1)); obs_e = take(no * (N + 1));
        rt = take(m); rnu = take(m); rdt = take(m); rdnu = take(m); rc = take(m); rd = take(m); rtt = take(m);
        Hd = take((N + 1) * D::NZ); hg = take((N + 1) * D::NZ); ga = take((N + 1) * D::NZ);
        Jc = take(N * no * D::NZ); wJ = take(N * no); kS = take(N); kE = take(N); kC = take(N); gC = take(N);
        P = take(D::NX * D::NX); pv = take(D::NX); T = take(D::NX * D::NZ); H = take(D::NZ * D::NZ);
        hv = take(D::NZ);
        Kk = take(N * D::NU * D::NX); kf = take(N * D::NU); lam = take(D::NZ);
        Fth = take(MAXF); Fph = take(MAXF); cst = take(16);
    }
};

// ------------------------------------------------------------------------------------------------
// problem context shared by the device functions of one solve
// ------------------------------------------------------------------------------------------------
template <int NOBS>
struct Ctx {
    using D = Dim<NOBS>;
    Lds<NOBS> s;
    int N, lane, nobs;      // nobs = obstacles actually present in this problem (<= NOBS)
    double lin_sN, cconst, wsig;
    double alpha, om, cm, Ls, Ws;
    int degree;
};

// CBF pieces between stages i and i+1 for obstacle o, at the point (x + al*dx)
template <int NOBS>
__device__ __forceinline__ void cbf_terms(const Ctx<NOBS>& c, int o, int i, double al, double& dsc,
                                          double& dec, double& dsn, double& den) {
    using D = Dim<NOBS>;
    const double* x = c.s.x;
    const double* dx = c.s.dx;
    const int N1 = c.N + 1;
    double sc = x[i * 6 + 4] + al * dx[i * D::NX + 4], ec = x[i * 6 + 5] + al * dx[i * D::NX + 5];
    double sn = x[(i + 1) * 6 + 4] + al * dx[(i + 1) * D::NX + 4];
    double en = x[(i + 1) * 6 + 5] + al * dx[(i + 1) * D::NX + 5];
    dsc = (sc - c.s.obs_s[o * N1 + i] - c.s.cst[12 + o]) / c.Ls;   // lap-corrected   (control.py:539-540)
    dec = (ec - c.s.obs_e[o * N1 + i]) / c.Ws;
    dsn = (sn - c.s.obs_s[o * N1 + i + 1]) / c.Ls;              // NOT corrected   (control.py:542, quirk Q1)
    den = (en - c.s.obs_e[o * N1 + i + 1]) / c.Ws;
}

// value of row j (unscaled) at the point (x,u,sg) + al*(dx,du)
template <int NOBS>
__device__ __forceinline__ double row_value(const Ctx<NOBS>& c, int j, double al) {
    using D = Dim<NOBS>;
    const int N = c.N;
    if (j >= N * D::NR) {  // sigma_0^o >= 0
        int o = j - N * D::NR;
        return c.s.sg[o] + al * c.s.dx[6 + o];
    }
    int k = j / D::NR, r = j - k * D::NR;
    if (r < 4) {
        int i = r >> 1;
        double v = c.s.u[k * 2 + i] + al * c.s.du[k * D::NU + i];
        return (r & 1) ? c.s.cst[10 + i] - v : v - c.s.cst[8 + i];
    }
    if (r < 8) {
        int comp = (r < 6) ? 0 : 5;
        double v = c.s.x[(k + 1) * 6 + comp] + al * c.s.dx[(k + 1) * D::NX + comp];
        double lo = (r < 6) ? c.s.vlo[k + 1] : c.s.elo[k + 1];
        double hi = (r < 6) ? c.s.vhi[k + 1] : c.s.ehi[k + 1];
        return (r & 1) ? hi - v : v - lo;
    }
    if (r < 8 + NOBS) {
        int o = r - 8;
        return c.s.sg[(k + 1) * (NOBS ? NOBS : 1) + o] + al * c.s.du[k * D::NU + 2 + o];
    }
    int o = r - 8 - NOBS;
    double dsc, dec, dsn, den;
    cbf_terms(c, o, k, al, dsc, dec, dsn, den);
    int q = c.degree;
    double gc = ipow_d(dsc, q) + ipow_d(dec, q), gn = ipow_d(dsn, q) + ipow_d(den, q);
    const int no = NOBS ? NOBS : 1;
    double sk = c.s.sg[k * no + o] + al * c.s.dx[k * D::NX + 6 + o];
    double sk1 = c.s.sg[(k + 1) * no + o] + al * c.s.du[k * D::NU + 2 + o];
    return gn - sk1 - c.om * (gc - sk) - c.alpha * c.cm;
}

// cost at (x,u,sg) + al*(dx,du); every lane returns the total
template <int NOBS>
__device__ __forceinline__ double cost_value(const Ctx<NOBS>& c, double al) {
    using D = Dim<NOBS>;
    const int N = c.N, no = NOBS ? NOBS : 1;
    double acc = 0.0;
    for (int e = c.lane; e < (N + 1) * 6; e += WAVE) {
        int k = e / 6, i = e - k * 6;
        double v = c.s.x[e] + al * c.s.dx[k * D::NX + i];
        double d = v - c.s.xr[e];
        acc += c.s.cst[i] * d * d;
        if (k == N && i == 4) acc += c.lin_sN * v;
    }
    for (int e = c.lane; e < N * 2; e += WAVE) {
        int k = e >> 1, i = e & 1;
        double v = c.s.u[e] + al * c.s.du[k * D::NU + i];
        acc += c.s.cst[6 + i] * v * v;
    }
    for (int k = c.lane; k < N; k += WAVE) {
        double e1 = c.s.x[(k + 1) * 6 + 5] + al * c.s.dx[(k + 1) * D::NX + 5];
        double e0 = c.s.x[k * 6 + 5] + al * c.s.dx[k * D::NX + 5];
        acc += c.s.wc[k] * (e1 - e0) * (e1 - e0);
    }
    if (NOBS) {
        for (int e = c.lane; e < (N + 1) * NOBS; e += WAVE) {
            int k = e / no, o = e - k * no;
            if (o < c.nobs) {
                double v = c.s.sg[e] + al * (k == 0 ? c.s.dx[6 + o] : c.s.du[(k - 1) * D::NU + 2 + o]);
                acc += c.wsig * v;
            }
        }
    }
    return wave_sum(acc) + c.cconst;
}

// exact directional derivative of the cost along (dx,du): d/da f(z + a dz) at a = 0
template <int NOBS>
__device__ __forceinline__ double cost_dir(const Ctx<NOBS>& c) {
    using D = Dim<NOBS>;
    const int N = c.N, no = NOBS ? NOBS : 1;
    double acc = 0.0;
    for (int e = c.lane; e < (N + 1) * 6; e += WAVE) {
        int k = e / 6, i = e - k * 6;
        double dv = c.s.dx[k * D::NX + i];
        acc += 2.0 * c.s.cst[i] * (c.s.x[e] - c.s.xr[e]) * dv;
        if (k == N && i == 4) acc += c.lin_sN * dv;
    }
    for (int e = c.lane; e < N * 2; e += WAVE) {
        int k = e >> 1, i = e & 1;
        acc += 2.0 * c.s.cst[6 + i] * c.s.u[e] * c.s.du[k * D::NU + i];
    }
    for (int k = c.lane; k < N; k += WAVE) {
        double de = c.s.x[(k + 1) * 6 + 5] - c.s.x[k * 6 + 5];
        double dd = c.s.dx[(k + 1) * D::NX + 5] - c.s.dx[k * D::NX + 5];
        acc += 2.0 * c.s.wc[k] * de * dd;
    }
    if (NOBS) {
        for (int e = c.lane; e < (N + 1) * NOBS; e += WAVE) {
            int k = e / no, o = e - k * no;
            if (o < c.nobs) acc += c.wsig * (k == 0 ? c.s.dx[6 + o] : c.s.du[(k - 1) * D::NU + 2 + o]);
        }
    }
    return wave_sum(acc);
}

// is row j part of the problem?  (infinite bounds and absent obstacles are skipped)
template <int NOBS>
__device__ __forceinline__ bool row_active(const Ctx<NOBS>& c, int j) {
    using D = Dim<NOBS>;
    const int N = c.N;
    if (j >= N * D::NR) return (j - N * D::NR) < c.nobs;
    int k = j / D::NR, r = j - k * D::NR;
    if (r < 4) return true;
    if (r < 8) {
        double b = (r == 4) ? c.s.vlo[k + 1] : (r == 5) ? c.s.vhi[k + 1] : (r == 6) ? c.s.elo[k + 1] : c.s.ehi[k + 1];
        return isfinite(b);
    }
    if (r < 8 + NOBS) return (r - 8) < c.nobs;
    return (r - 8 - NOBS) < c.nobs;
}

// ------------------------------------------------------------------------------------------------
// Jacobian-dependent assembly: ga (Lagrangian gradient per stage, for the KKT error), Jc (CBF row
// Jacobians in stage coordinates).  Needs x,u,sg and rnu.
// stage coordinates z_k = [x_k (6), sigma_k (NOBS), u_k (2), sigma_{k+1} (NOBS)]
// ------------------------------------------------------------------------------------------------
template <int NOBS>
__device__ __forceinline__ void assemble_first_order(Ctx<NOBS>& c) {
    using D = Dim<NOBS>;
    const int N = c.N, no = NOBS ? NOBS : 1;
    // CBF Jacobians
    if (NOBS) {
        for (int e = c.lane; e < N * NOBS; e += WAVE) {
            int k = e / NOBS, o = e - k * NOBS;
            double* J = c.s.Jc + (k * no + o) * D::NZ;
            if (o < c.nobs) {
                double dsc, dec, dsn, den;
                cbf_terms(c, o, k, 0.0, dsc, dec, dsn, den);
                int q = c.degree;
                double gsn = q * ipow_d(dsn, q - 1) / c.Ls, gen = q * ipow_d(den, q - 1) / c.Ws;
                double gsc = q * ipow_d(dsc, q - 1) / c.Ls, gec = q * ipow_d(dec, q - 1) / c.Ws;
                double d = c.s.rd[k * D::NR + 8 + NOBS + o];
                for (int a = 0; a < D::NZ; a++)
                    J[a] = d * (gsn * c.s.M[4 * D::NZ + a] + gen * c.s.M[5 * D::NZ + a]);
                J[4] -= d * c.om * gsc;
                J[5] -= d * c.om * gec;
                J[6 + o] += d * c.om;
                J[D::NX + 2 + o] -= d;
            } else {
                for (int a = 0; a < D::NZ; a++) J[a] = 0.0;
            }
        }
    }
    SYNC();
    // ga[k][a]: gradient of the Lagrangian f - nu'c with respect to stage coordinates
    for (int e = c.lane; e < (N + 1) * D::NZ; e += WAVE) {
        int k = e / D::NZ, a = e - k * D::NZ;
        double g = 0.0;
        if (a < 6) {  // x_k component
            g = 2.0 * c.s.cst[a] * (c.s.x[k * 6 + a] - c.s.xr[k * 6 + a]);
            if (k == N && a == 4) g += c.lin_sN;
            if (k >= 1 && (a == 0 || a == 5)) {  // box rows of x_k live in stage k-1 slots 4..7
                int j0 = (k - 1) * D::NR + (a == 0 ? 4 : 6);
                if (row_active(c, j0)) g -= c.s.rnu[j0];
                if (row_active(c, j0 + 1)) g += c.s.rnu[j0 + 1];
            }
        } else if (a < D::NX) {  // sigma_k as a state: only sigma_0 carries its own terms here
            int o = a - 6;
            if (k == 0 && o < c.nobs) g = c.wsig - c.s.rnu[N * D::NR + o];
        } else if (k < N) {
            if (a < D::NX + 2) {
                int i = a - D::NX;
                g = 2.0 * c.s.cst[6 + i] * c.s.u[k * 2 + i] - c.s.rnu[k * D::NR + 2 * i] + c.s.rnu[k * D::NR + 2 * i + 1];
            } else {
                int o = a - D::NX - 2;
                if (o < c.nobs) g = c.wsig - c.s.rnu[k * D::NR + 8 + o];
            }
        }
        if (k < N) {
            // coupling cost wc_k (ey_{k+1} - ey_k)^2 in stage coordinates: mc = M[5,:] - e_ey
            double wck = c.s.wc[k];
            if (wck != 0.0) {
                double de = c.s.x[(k + 1) * 6 + 5] - c.s.x[k * 6 + 5];
                double mc = c.s.M[5 * D::NZ + a] - (a == 5 ? 1.0 : 0.0);
                g += 2.0 * wck * de * mc;
            }
            if (NOBS)
                for (int o = 0; o < c.nobs; o++)
                    g -= c.s.rnu[k * D::NR + 8 + NOBS + o] * c.s.Jc[(k * no + o) * D::NZ + a];
        }
        if (k == N && a >= D::NX) g = 0.0;
        c.s.ga[e] = g;
    }
    SYNC();
}

// adjoint sweep: infinity norm of the reduced Lagrangian gradient (inputs of every stage + sigma_0).
// Side effect: ga is overwritten with the SAME gradient in "reduced form" -- the costates lam_k of
// the sweep are folded into the stage gradients (adding lam'(M dz_k - dx_{k+1}) = 0 to the Newton
// QP), so that the state part vanishes identically and the input part is the (small) reduced
// gradient.  Feeding the Riccati recursion this form instead of the raw stage gradients keeps the
// O(nu) terms that cancel near the solution out of the vector recursion, so the Newton step's
// rounding error is relative to the residual rather than to |nu|.
template <int NOBS>
__device__ __forceinline__ double dual_infeasibility(Ctx<NOBS>& c) {
    using D = Dim<NOBS>;
    const int N = c.N;
    double emax = 0.0;
    if (c.lane < D::NX) c.s.lam[c.lane] = c.s.ga[N * D::NZ + c.lane];
    SYNC();
    if (c.lane < D::NZ) c.s.ga[N * D::NZ + c.lane] = 0.0;
    for (int k = N - 1; k >= 0; k--) {
        double tot = 0.0;
        if (c.lane < D::NZ) {
            tot = c.s.ga[k * D::NZ + c.lane];
            for (int i = 0; i < D::NX; i++) tot += c.s.M[i * D::NZ + c.lane] * c.s.lam[i];
            if (c.lane >= D::NX) emax = fmax(emax, fabs(tot));
            bool keep = c.lane >= D::NX || (k == 0 && c.lane >= 6);
            c.s.ga[k * D::NZ + c.lane] = keep ? tot : 0.0;
        }
        SYNC();
        if (c.lane < D::NX) c.s.lam[c.lane] = tot;
        SYNC();
    }
    if (NOBS && c.lane >= 6 && c.lane < 6 + c.nobs) emax = fmax(emax, fabs(c.s.lam[c.lane]));
    return wave_max(emax);
}

// Second-order / barrier assembly for the Newton system at barrier parameter mu.
//   Hd  diagonal of the stage Hessian (cost + Sigma of simple rows + "current" CBF curvature)
//   hg  stage gradient of the barrier-modified objective:  grad f - J'(mu/t - Sigma*rp)
//   wJ  Sigma of the CBF rows, kS/kE "next" CBF curvature (added to P[4][4], P[5][5]),
//   kC/gC coupling cost second/first-order coefficients
template <int NOBS>
__device__ __forceinline__ void assemble_newton(Ctx<NOBS>& c, double mu) {
    using D = Dim<NOBS>;
    const int N = c.N, no = NOBS ? NOBS : 1;
    for (int e = c.lane; e < (N + 1) * D::NZ; e += WAVE) {
        int k = e / D::NZ, a = e - k * D::NZ;
        double h = 0.0, g = c.s.ga[e];  // g starts from the reduced-form Lagrangian gradient
        auto add_row = [&](int j, double sign) {  // simple row with Jacobian sign*e_a
            if (!row_active(c, j)) return;
            double t = c.s.rt[j], nu = c.s.rnu[j];
            double sig = nu / t, rp = c.s.rc[j] - t;
            h += sig;
            g += sign * (nu - mu / t + sig * rp);
        };
        if (a < 6) {
            h = 2.0 * c.s.cst[a];
            if (k >= 1 && (a == 0 || a == 5)) {
                int j0 = (k - 1) * D::NR + (a == 0 ? 4 : 6);
                add_row(j0, 1.0);
                add_row(j0 + 1, -1.0);
            }
        } else if (a < D::NX) {
            int o = a - 6;
            if (k == 0 && o < c.nobs) {
                add_row(N * D::NR + o, 1.0);
            } else if (k == 0) {
                h = 1.0;  // absent obstacle: pin its sigma_0
            }
        } else if (k < N) {
            if (a < D::NX + 2) {
                int i = a - D::NX;
                h = 2.0 * c.s.cst[6 + i];
                add_row(k * D::NR + 2 * i, 1.0);
                add_row(k * D::NR + 2 * i + 1, -1.0);
            } else {
                int o = a - D::NX - 2;
                if (o < c.nobs) {
                    add_row(k * D::NR + 8 + o, 1.0);
                } else {
                    h = 1.0;  // absent obstacle: pin sigma_{k+1}
                }
            }
        }
        if (k < N && NOBS) {
            for (int o = 0; o < c.nobs; o++) {
                int j = k * D::NR + 8 + NOBS + o;
                double t = c.s.rt[j], nu = c.s.rnu[j];
                double sig = nu / t, rp = c.s.rc[j] - t;
                g += c.s.Jc[(k * no + o) * D::NZ + a] * (nu - mu / t + sig * rp);
            }
        }
        if (k == N && a >= D::NX) { h = 0.0; g = 0.0; }
        c.s.Hd[e] = h;
        c.s.hg[e] = g;
    }
    for (int k = c.lane; k < N; k += WAVE) {
        double ks = 0.0, ke = 0.0;
        if (NOBS) {
            for (int o = 0; o < c.nobs; o++) {
                int j = k * D::NR + 8 + NOBS + o;
                double t = c.s.rt[j], nu = c.s.rnu[j], d = c.s.rd[j];
                c.s.wJ[k * no + o] = nu / t;
                double dsc, dec, dsn, den;
                cbf_terms(c, o, k, 0.0, dsc, dec, dsn, den);
                int q = c.degree;
                double qq = (double)(q * (q - 1));
                ks -= nu * d * qq * ipow_d(dsn, q - 2) / (c.Ls * c.Ls);
                ke -= nu * d * qq * ipow_d(den, q - 2) / (c.Ws * c.Ws);
            }
        }
        c.s.kS[k] = ks;
        c.s.kE[k] = ke;
        double wck = c.s.wc[k];
        c.s.kC[k] = 2.0 * wck;
        c.s.gC[k] = 2.0 * wck * (c.s.x[(k + 1) * 6 + 5] - c.s.x[k * 6 + 5]);
    }
    SYNC();
    // "current" CBF curvature lands on the diagonal of (s_k, ey_k); coupling gradient on hg
    for (int e = c.lane; e < N * D::NZ; e += WAVE) {
        int k = e / D::NZ, a = e - k * D::NZ;
        double h = 0.0;
        if (NOBS && (a == 4 || a == 5)) {
            for (int o = 0; o < c.nobs; o++) {
                int j = k * D::NR + 8 + NOBS + o;
                double nu = c.s.rnu[j], d = c.s.rd[j];
                double dsc, dec, dsn, den;
                cbf_terms(c, o, k, 0.0, dsc, dec, dsn, den);
                int q = c.degree;
                double qq = (double)(q * (q - 1));
                h += (a == 4) ? nu * d * c.om * qq * ipow_d(dsc, q - 2) / (c.Ls * c.Ls)
                              : nu * d * c.om * qq * ipow_d(dec, q - 2) / (c.Ws * c.Ws);
            }
        }
        c.s.Hd[e] += h;
    }
    SYNC();
}

// ------------------------------------------------------------------------------------------------
// Riccati backward sweep with regularisation dw on the input / sigma_0 diagonal.
// Returns false if a pivot is not positive (wrong inertia).  On success Kk/kf hold the feedback
// and dx[0] the step of the free initial components (sigma_0).
// ------------------------------------------------------------------------------------------------
template <int NOBS>
__device__ __forceinline__ bool riccati_backward(Ctx<NOBS>& c, double dw) {
    using D = Dim<NOBS>;
    constexpr int NX = D::NX, NU = D::NU, NZ = D::NZ;
    const int N = c.N, lane = c.lane, no = NOBS ? NOBS : 1;
    // terminal: P_N = diag(Hd[N][0..NX)) (+ stage N-1 extras on (s,ey)), p_N = hg[N]
    for (int e = lane; e < NX * NX; e += WAVE) {
        int i = e / NX, j = e - i * NX;
        double v = 0.0;
        if (i == j) {
            v = c.s.Hd[N * NZ + i];
            if (i == 4) v += c.s.kS[N - 1];
            if (i == 5) v += c.s.kE[N - 1] + c.s.kC[N - 1];
        }
        c.s.P[e] = v;
    }
    if (lane < NX) c.s.pv[lane] = c.s.hg[N * NZ + lane];
    SYNC();
    bool ok = true;
    for (int k = N - 1; k >= 0; k--) {
        // T = P M
        for (int e = lane; e < NX * NZ; e += WAVE) {
            int i = e / NZ, a = e - i * NZ;
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < NX; j++) s += c.s.P[i * NX + j] * c.s.M[j * NZ + a];
            c.s.T[e] = s;
        }
        SYNC();
        // H = M'T + stage terms ; hv = M'p + hg
        const double kc = c.s.kC[k];
        for (int e = lane; e < NZ * NZ; e += WAVE) {
            int r = e / NZ, a = e - r * NZ;
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NX; i++) s += c.s.M[i * NZ + r] * c.s.T[i * NZ + a];
            if (r == a) {
                s += c.s.Hd[k * NZ + r];
                if (r >= NX || (k == 0 && r >= 6)) s += dw;
            }
            if (NOBS)
                for (int o = 0; o < c.nobs; o++) {
                    const double* J = c.s.Jc + (k * no + o) * NZ;
                    s += c.s.wJ[k * no + o] * J[r] * J[a];
                }
            if (kc != 0.0) {
                // kC (m_e - e_ey)(m_e - e_ey)' with the m_e m_e' part already inside P[5][5]
                if (r == 5) s -= kc * c.s.M[5 * NZ + a];
                if (a == 5) s -= kc * c.s.M[5 * NZ + r];
                if (r == 5 && a == 5) s += kc;
            }
            c.s.H[e] = s;
        }
        if (lane < NZ) {
            double s = c.s.hg[k * NZ + lane];
#pragma unroll
            for (int i = 0; i < NX; i++) s += c.s.M[i * NZ + lane] * c.s.pv[i];
            c.s.hv[lane] = s;
        }
        SYNC();
        // every lane factorises Huu = L D L' itself (NU <= 5, broadcast LDS reads).  Unit-lower L,
        // pivots D, reciprocal pivots rD: no square roots, and one division per pivot on the
        // dependent chain (FP64 sqrt/div are ~20-instruction sequences on CDNA4).
        double L[NU][NU], Dp[NU], rD[NU];
#pragma unroll
        for (int a = 0; a < NU; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) L[a][b] = c.s.H[(NX + a) * NZ + NX + b];
#pragma unroll
        for (int j = 0; j < NU; j++) {
            double d = L[j][j];
#pragma unroll
            for (int q = 0; q < j; q++) d -= L[j][q] * L[j][q] * Dp[q];
            if (!(d > 0.0)) ok = false;
            Dp[j] = d;
            rD[j] = 1.0 / d;
#pragma unroll
            for (int i = j + 1; i < NU; i++) {
                double t = L[i][j];
#pragma unroll
                for (int q = 0; q < j; q++) t -= L[i][q] * L[j][q] * Dp[q];
                L[i][j] = t * rD[j];
            }
        }
        if (!ok) break;  // uniform: every lane computed the same pivots
        // Factorised update, the block-Cholesky form of the recursion (Huu = L D L'):
        //   Y = L^{-1} Hux  (unit forward substitution),  P_new = Hxx - Y' D^{-1} Y,  K = -L^{-T} D^{-1} Y,
        //   yg = L^{-1} gu,  p_new = gx - Y' D^{-1} yg,   kff = -L^{-T} D^{-1} yg.
        // P_new is formed symmetrically from the SAME factor for (i,j) and (j,i), which keeps it
        // symmetric positive semi-definite under barrier weights Sigma up to ~1e13 (the K-form
        // Hxx + Hux'K does not).  Column index NX stands for the gradient column.
        constexpr int PCNT = (NX * (NX + 1) + WAVE - 1) / WAVE;
        double Pn[PCNT];
#pragma unroll
        for (int cnt = 0; cnt < PCNT; cnt++) {
            const int e = lane + cnt * WAVE;
            if (e >= NX * (NX + 1)) { Pn[cnt] = 0.0; continue; }
            int i = e / (NX + 1), j = e - i * (NX + 1);
            double yi[NU], yj[NU];
#pragma unroll
            for (int a = 0; a < NU; a++) {
                yi[a] = c.s.H[(NX + a) * NZ + i];
                yj[a] = (j < NX) ? c.s.H[(NX + a) * NZ + j] : c.s.hv[NX + a];
            }
#pragma unroll
            for (int a = 1; a < NU; a++) {
#pragma unroll
                for (int q = 0; q < a; q++) { yi[a] -= L[a][q] * yi[q]; yj[a] -= L[a][q] * yj[q]; }
            }
            double s = (j < NX) ? c.s.H[i * NZ + j] : c.s.hv[i];
#pragma unroll
            for (int a = 0; a < NU; a++) { yj[a] *= rD[a]; s -= yi[a] * yj[a]; }
            Pn[cnt] = s;
            if (i == 0) {
                // feedback column j: K[:,j] = -L^{-T} (D^{-1} yj)
#pragma unroll
                for (int a = NU - 2; a >= 0; a--) {
#pragma unroll
                    for (int q = a + 1; q < NU; q++) yj[a] -= L[q][a] * yj[q];
                }
#pragma unroll
                for (int a = 0; a < NU; a++) {
                    if (j < NX) c.s.Kk[(k * NU + a) * NX + j] = -yj[a];
                    else c.s.kf[k * NU + a] = -yj[a];
                }
            }
        }
        SYNC();
#pragma unroll
        for (int cnt = 0; cnt < PCNT; cnt++) {
            const int e = lane + cnt * WAVE;
            if (e >= NX * (NX + 1)) continue;
            int i = e / (NX + 1), j = e - i * (NX + 1);
            double v = Pn[cnt];
            if (j < NX) {
                if (k >= 1 && i == j) {  // stage k-1 extras on (s_k, ey_k)
                    if (i == 4) v += c.s.kS[k - 1];
                    if (i == 5) v += c.s.kE[k - 1] + c.s.kC[k - 1];
                }
                c.s.P[i * NX + j] = v;
            } else {
                c.s.pv[i] = v;
            }
        }
        SYNC();
    }
    if (!ok) { SYNC(); return false; }
    // free initial components sigma_0: minimise 1/2 d'P d + p'd over them (x_0 is fixed)
    if (lane < NX) c.s.dx[lane] = 0.0;
    if (NOBS) {
        double L[NOBS ? NOBS : 1][NOBS ? NOBS : 1];
        double y[NOBS ? NOBS : 1];
#pragma unroll
        for (int a = 0; a < NOBS; a++) {
#pragma unroll
            for (int b = 0; b <= a; b++) L[a][b] = c.s.P[(6 + a) * NX + 6 + b];
            y[a] = -c.s.pv[6 + a];
        }
#pragma unroll
        for (int j = 0; j < NOBS; j++) {
            double d = L[j][j];
#pragma unroll
            for (int q = 0; q < j; q++) d -= L[j][q] * L[j][q];
            if (!(d > 0.0)) ok = false;
            d = sqrt(d);
            L[j][j] = d;
#pragma unroll
            for (int i = j + 1; i < NOBS; i++) {
                double t = L[i][j];
#pragma unroll
                for (int q = 0; q < j; q++) t -= L[i][q] * L[j][q];
                L[i][j] = t / d;
            }
        }
        if (ok) {
#pragma unroll
            for (int a = 0; a < NOBS; a++) {
#pragma unroll
                for (int q = 0; q < a; q++) y[a] -= L[a][q] * y[q];
                y[a] /= L[a][a];
            }
#pragma unroll
            for (int a = NOBS - 1; a >= 0; a--) {
#pragma unroll
                for (int q = a + 1; q < NOBS; q++) y[a] -= L[q][a] * y[q];
                y[a] /= L[a][a];
            }
            SYNC();
            if (lane == 0) {
#pragma unroll
                for (int a = 0; a < NOBS; a++) c.s.dx[6 + a] = y[a];
            }
        }
    }
    SYNC();
    return ok;
}

template <int NOBS>
__device__ __forceinline__ void riccati_forward(Ctx<NOBS>& c) {
    using D = Dim<NOBS>;
    constexpr int NX = D::NX, NU = D::NU, NZ = D::NZ;
    const int N = c.N, lane = c.lane;
    for (int k = 0; k < N; k++) {
        if (lane < NU) {
            double s = c.s.kf[k * NU + lane];
#pragma unroll
            for (int j = 0; j < NX; j++) s += c.s.Kk[(k * NU + lane) * NX + j] * c.s.dx[k * NX + j];
            c.s.du[k * NU + lane] = s;
        }
        SYNC();
        if (lane < NX) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < NX; j++) s += c.s.M[lane * NZ + j] * c.s.dx[k * NX + j];
#pragma unroll
            for (int a = 0; a < NU; a++) s += c.s.M[lane * NZ + NX + a] * c.s.du[k * NU + a];
            c.s.dx[(k + 1) * NX + lane] = s;
        }
        SYNC();
    }
}

// ------------------------------------------------------------------------------------------------
// the solver
// ------------------------------------------------------------------------------------------------
template <int NOBS>
__global__ void __launch_bounds__(WAVE) crx_solve_kernel(const crx_kparams kp) {
    using D = Dim<NOBS>;
    constexpr int NX = D::NX, NU = D::NU, NZ = D::NZ, NR = D::NR;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x, lane = threadIdx.x, N = kp.N, no = NOBS ? NOBS : 1;
    if (b >= kp.batch) return;
    Ctx<NOBS> c;
    c.s.carve(smem, N);
    c.N = N; c.lane = lane;
    const int m = c.s.m;

    // ---- load: one coalesced pass over this problem's inputs -------------------------------------
    for (int e = lane; e < NX * NZ; e += WAVE) {
        int i = e / NZ, a = e - i * NZ;
        double v = 0.0;
        if (i < 6) {
            if (a < 6) v = kp.A[i * 6 + a];
            else if (a >= NX && a < NX + 2) v = kp.B[i * 2 + (a - NX)];
        } else if (a == NX + 2 + (i - 6)) v = 1.0;
        c.s.M[e] = v;
    }
    if (lane < 6) c.s.x[lane] = kp.x0[(size_t)b * 6 + lane];
    if (lane == 0) {
        for (int i = 0; i < 6; i++) c.s.cst[i] = kp.wq[i];
        c.s.cst[6] = kp.wr[0]; c.s.cst[7] = kp.wr[1];
        c.s.cst[8] = -kp.delta_max; c.s.cst[10] = kp.delta_max; c.s.cst[9] = -kp.a_max; c.s.cst[11] = kp.a_max;
        c.s.cst[12] = c.s.cst[13] = c.s.cst[14] = 0.0;
    }
    c.alpha = kp.alpha; c.om = 1.0 - kp.alpha; c.cm = 1.0 + kp.margin; c.Ls = kp.l_sum; c.Ws = kp.w_sum;
    c.degree = kp.degree; c.wsig = kp.w_slack; c.lin_sN = 0.0; c.cconst = 0.0;
    c.nobs = 0;
    int infeas0 = 0;
    SYNC();
    if (kp.mode == 0) {
        // planner front-end (overtake_traj_planner.py:266-334); bez arrays staged through Fth.. scratch
        double* bs = c.s.dx;               // scratch: N+1 values
        double* be = c.s.dx + (N + 1);
        for (int j = lane; j <= N; j += WAVE) {
            bs[j] = kp.bez_s[(size_t)b * (N + 1) + j];
            be[j] = kp.bez_ey[(size_t)b * (N + 1) + j];
        }
        SYNC();
        const double s0 = c.s.x[4], vx0 = c.s.x[0];
        for (int j = lane; j <= N; j += WAVE) {
            double st = s0 + 1.0 * j * vx0 * kp.dt_ref;                             // :330
            st = fmin(fmax(st, bs[0]), bs[N]);                                       // :331
            for (int i = 0; i < 6; i++) c.s.xr[j * 6 + i] = 0.0;
            c.s.xr[j * 6 + 4] = st;
            c.s.xr[j * 6 + 5] = interp_lin(bs, be, N + 1, st);                       // :332
            c.s.vlo[j] = -INFINITY;
            c.s.vhi[j] = j >= 1 ? kp.v_max : INFINITY;                               // :276
            c.s.elo[j] = j < N ? kp.ey_lb[(size_t)b * N + j] : -INFINITY;            // :277-324
            c.s.ehi[j] = j < N ? kp.ey_ub[b] : INFINITY;
            if (j < N) c.s.wc[j] = (j >= 1 && j <= N - 2) ? kp.w_dey : 0.0;          // :325-327
        }
        c.lin_sN = -kp.w_prog;                                                       // :328
        c.cconst = kp.w_prog * s0;
        SYNC();
        if (c.s.x[5] < c.s.elo[0] - kp.opts.tol || c.s.x[5] > c.s.ehi[0] + kp.opts.tol) infeas0 = 1;
    } else {
        c.nobs = kp.n_obs ? kp.n_obs[b] : NOBS;
        for (int e = lane; e < (N + 1) * 6; e += WAVE)
            c.s.xr[e] = kp.per_stage_target ? kp.xt[(size_t)b * (N + 1) * 6 + e] : kp.xt[(size_t)b * 6 + (e % 6)];
        for (int j = lane; j <= N; j += WAVE) {
            c.s.vlo[j] = kp.v_min; c.s.vhi[j] = kp.v_max; c.s.elo[j] = -kp.ey_max; c.s.ehi[j] = kp.ey_max;
            if (j < N) c.s.wc[j] = 0.0;
        }
        if (NOBS) {
            for (int e = lane; e < NOBS * (N + 1); e += WAVE) {
                int o = e / (N + 1);
                bool on = o < c.nobs;
                c.s.obs_s[e] = on ? kp.obs_s[((size_t)b * kp.n_obs_max) * (N + 1) + e] : 0.0;
                c.s.obs_e[e] = on ? kp.obs_ey[((size_t)b * kp.n_obs_max) * (N + 1) + e] : 0.0;
            }
            if (lane < NOBS) c.s.cst[12 + lane] = lane < c.nobs ? kp.lap_off[(size_t)b * kp.n_obs_max + lane] : 0.0;
        }
        SYNC();
        if (c.s.x[0] < kp.v_min - kp.opts.tol || c.s.x[0] > kp.v_max + kp.opts.tol ||
            c.s.x[5] < -kp.ey_max - kp.opts.tol || c.s.x[5] > kp.ey_max + kp.opts.tol)
            infeas0 = 1;                                                             // quirk Q9
    }

    // ---- starting point: u = 0, sigma = 0, x by roll-out ----------------------------------------
    for (int e = lane; e < N * 2; e += WAVE) c.s.u[e] = 0.0;
    for (int e = lane; e < (N + 1) * no; e += WAVE) c.s.sg[e] = 0.0;
    for (int e = lane; e < (N + 1) * NX; e += WAVE) c.s.dx[e] = 0.0;
    for (int e = lane; e < N * NU; e += WAVE) c.s.du[e] = 0.0;
    SYNC();
    for (int k = 0; k < N; k++) {
        if (lane < 6) {
            double s = 0.0;
            for (int j = 0; j < 6; j++) s += c.s.M[lane * NZ + j] * c.s.x[k * 6 + j];
            c.s.x[(k + 1) * 6 + lane] = s;
        }
        SYNC();
    }
    // row scaling (CBF rows only), slack and multiplier start
    for (int j = lane; j < m; j += WAVE) {
        double d = 1.0;
        if (NOBS && j < N * NR) {
            int k = j / NR, r = j - k * NR;
            if (r >= 8 + NOBS && (r - 8 - NOBS) < c.nobs) {
                int o = r - 8 - NOBS, q = c.degree;
                double dsc, dec, dsn, den;
                cbf_terms(c, o, k, 0.0, dsc, dec, dsn, den);
                double gm = 1.0;
                gm = fmax(gm, fabs(q * ipow_d(dsn, q - 1) / c.Ls));
                gm = fmax(gm, fabs(q * ipow_d(den, q - 1) / c.Ws));
                if (k > 0) {
                    gm = fmax(gm, fabs(c.om * q * ipow_d(dsc, q - 1) / c.Ls));
                    gm = fmax(gm, fabs(c.om * q * ipow_d(dec, q - 1) / c.Ws));
                }
                d = fmin(1.0, kp.opts.grad_scale_max / gm);
            }
        }
        c.s.rd[j] = d;
        c.s.rnu[j] = 1.0;
    }
    SYNC();
    for (int j = lane; j < m; j += WAVE) {
        bool act = row_active(c, j);
        double cv = act ? c.s.rd[j] * row_value(c, j, 0.0) : 1.0;
        c.s.rc[j] = cv;
        c.s.rt[j] = act ? fmax(fabs(cv), kp.opts.slack_push) : 1.0;
        if (!act) c.s.rnu[j] = 0.0;
    }
    SYNC();
    assemble_first_order(c);
    // multiplier start on simple-bound rows: the cost gradient that pushes against the bound.
    // With nu = 1 everywhere ga = grad f - (+-1), so grad f is recovered by adding the row terms back.
    {
        // reduced cost gradient by the adjoint sweep with nu = 0: reuse dual_infeasibility's recursion
        for (int j = lane; j < m; j += WAVE) c.s.rtt[j] = c.s.rnu[j];
        SYNC();
        for (int j = lane; j < m; j += WAVE) c.s.rnu[j] = 0.0;
        SYNC();
        assemble_first_order(c);
        if (lane < NX) c.s.lam[lane] = c.s.ga[N * NZ + lane];
        SYNC();
        for (int k = N - 1; k >= 0; k--) {
            double tot = 0.0;
            if (lane < NZ) {
                tot = c.s.ga[k * NZ + lane];
                for (int i = 0; i < NX; i++) tot += c.s.M[i * NZ + lane] * c.s.lam[i];
            }
            SYNC();
            if (lane < NX) c.s.lam[lane] = tot;
            if (lane >= NX && lane < NZ) c.s.hg[k * NZ + lane] = tot;  // scratch: reduced gradient of f
            SYNC();
        }
        for (int j = lane; j < m; j += WAVE) {
            double nu = c.s.rtt[j];
            if (nu != 0.0) {
                double gg = 0.0;
                if (j >= N * NR) gg = c.s.lam[6 + (j - N * NR)];
                else {
                    int k = j / NR, r = j - k * NR;
                    if (r < 4) gg = ((r & 1) ? -1.0 : 1.0) * c.s.hg[k * NZ + NX + (r >> 1)];
                    else if (r >= 8 && r < 8 + NOBS) gg = c.s.hg[k * NZ + NX + 2 + (r - 8)];
                }
                if (gg > 1.0) nu = gg;
            }
            c.s.rnu[j] = nu;
        }
        SYNC();
        assemble_first_order(c);
    }

    const crx_ipm_opts o = kp.opts;
    double mu = o.mu_init, dw_last = 0.0, E0 = INFINITY, theta_min = 0.0, theta_max = INFINITY;
    double f = cost_value(c, 0.0);
    int nf = 0, status = 1, it = 0;
    int mact = 0;
    for (int j = lane; j < m; j += WAVE) mact += row_active(c, j) ? 1 : 0;
    mact = (int)wave_sum((double)mact);
    const double kappa_sigma = 1e10, smax = 100.0, eta = 1e-8;

    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (it = 0;; it++) {
        long long tc0 = clock64();
        // ---- KKT error ---------------------------------------------------------------------------
        double nus = 0.0, e_p = 0.0, e_c = 0.0;
        for (int j = lane; j < m; j += WAVE) {
            if (!row_active(c, j)) continue;
            double t = c.s.rt[j], nu = c.s.rnu[j];
            nus += fabs(nu);
            e_p = fmax(e_p, fabs(c.s.rc[j] - t));
            e_c = fmax(e_c, fabs(t * nu));
        }
        nus = wave_sum(nus); e_p = wave_max(e_p); e_c = wave_max(e_c);
        double sd = fmax(smax, nus / (mact > 0 ? mact : 1)) / smax;
        long long tc1 = clock64();
        double e_d = dual_infeasibility(c) / sd;
        long long tc2 = clock64();
        e_c /= sd;
        E0 = fmax(e_d, fmax(e_p, e_c));
        if (E0 <= o.tol) { status = 0; break; }
        if (it >= o.max_iter) break;
        // ---- barrier update ----------------------------------------------------------------------
        for (;;) {
            double e_cm = 0.0;
            for (int j = lane; j < m; j += WAVE)
                if (row_active(c, j)) e_cm = fmax(e_cm, fabs(c.s.rt[j] * c.s.rnu[j] - mu));
            e_cm = wave_max(e_cm) / sd;
            double Emu = fmax(e_d, fmax(e_p, e_cm));
            if (Emu <= o.kappa_eps * mu && mu > o.tol / 10.0) {
                mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, o.theta_mu == 1.5 ? mu * sqrt(mu) : pow(mu, o.theta_mu)));
                nf = 0;
            } else
                break;
        }
        const double tau = fmax(o.tau_min, 1.0 - mu);
        // ---- Newton step -------------------------------------------------------------------------
        long long tc3 = clock64();
        assemble_newton(c, mu);
        long long tc4 = clock64();
        double dw = 0.0;
        bool ok = riccati_backward(c, 0.0);
        if (!ok) {
            dw = dw_last == 0.0 ? 1e-4 : fmax(1e-20, dw_last / 3.0);
            for (;;) {
                ok = riccati_backward(c, dw);
                if (ok) break;
                dw *= dw_last == 0.0 ? 100.0 : 8.0;
                if (dw > 1e40) break;
            }
            if (!ok) break;
            dw_last = dw;
        }
        long long tc5 = clock64();
        riccati_forward(c);
        long long tc6 = clock64();
        // ---- row steps, step lengths, merit pieces -------------------------------------------------
        double a_p = 1.0, a_d = 1.0, theta = 0.0, Dphi = 0.0, phi0 = 0.0;
        LogAcc lg0;
        for (int j = lane; j < m; j += WAVE) {
            if (!row_active(c, j)) { c.s.rdt[j] = 0.0; c.s.rdnu[j] = 0.0; continue; }
            double t = c.s.rt[j], nu = c.s.rnu[j], rp = c.s.rc[j] - t;
            // J dz: simple rows are linear -> difference of row values; CBF rows use Jc
            double jd;
            bool cbf = NOBS && j < N * NR && (j % NR) >= 8 + NOBS;
            if (cbf) {
                int k = j / NR, ob = (j % NR) - 8 - NOBS;
                const double* J = c.s.Jc + (k * no + ob) * NZ;
                jd = 0.0;
                for (int a = 0; a < NX; a++) jd += J[a] * c.s.dx[k * NX + a];
                for (int a = 0; a < NU; a++) jd += J[NX + a] * c.s.du[k * NU + a];
            } else {
                // simple rows: J dz straight from the step.  (Differencing row values instead loses
                // eps*|x| absolutely, which the update of nu amplifies by Sigma = nu/t ~ 1e10..1e13:
                // measured as a dual residual that stalls near 1e-6.)
                if (j >= N * NR) jd = c.s.dx[6 + (j - N * NR)];
                else {
                    int k = j / NR, r = j - k * NR;
                    if (r < 4) jd = ((r & 1) ? -1.0 : 1.0) * c.s.du[k * NU + (r >> 1)];
                    else if (r < 8) jd = ((r & 1) ? -1.0 : 1.0) * c.s.dx[(k + 1) * NX + (r < 6 ? 0 : 5)];
                    else jd = c.s.du[k * NU + 2 + (r - 8)];
                }
            }
            double dt = jd + rp;
            double dnu = (mu - t * nu - nu * dt) / t;
            c.s.rdt[j] = dt; c.s.rdnu[j] = dnu;
            if (dt < 0.0) a_p = fmin(a_p, -tau * t / dt);
            if (dnu < 0.0) a_d = fmin(a_d, -tau * nu / dnu);
            theta += fabs(rp);
            Dphi -= mu * dt / t;
            lg0.mul(t);
        }
        a_p = wave_min(a_p); a_d = wave_min(a_d); theta = wave_sum(theta);
        Dphi = wave_sum(Dphi) + cost_dir(c);
        phi0 = f - mu * lg0.wave_total();
        if (it == 0) {
            theta_min = 1e-4 * fmax(1.0, theta);
            theta_max = 1e4 * fmax(1.0, theta);
        }
        long long tc7 = clock64();
        // ---- filter line search --------------------------------------------------------------------
        double al = a_p, fn = f;
        int acc = 0, ftype = 0;
        // switching condition al * (-Dphi)^2.3 > theta^1.1 (only consulted when theta <= theta_min)
        const bool sw_try = (theta <= theta_min) && (Dphi < 0.0);
        const double sw_lhs = sw_try ? pow(-Dphi, 2.3) : 0.0, sw_rhs = sw_try ? pow(theta, 1.1) : 0.0;
        for (int ls = 0; ls < 40; ls++) {
            fn = cost_value(c, al);
            double phin = 0.0, thn = 0.0;
            LogAcc lg;
            for (int j = lane; j < m; j += WAVE) {
                if (!row_active(c, j)) continue;
                bool cbf = NOBS && j < N * NR && (j % NR) >= 8 + NOBS;
                double t = c.s.rt[j], dt = c.s.rdt[j];
                double cn = cbf ? c.s.rd[j] * row_value(c, j, al) : c.s.rc[j] + al * (dt - (c.s.rc[j] - t));
                double tn = t + al * dt;
                if (cn > tn) tn = cn;  // slack reset
                c.s.rtt[j] = tn;
                lg.mul(tn);
                thn += fabs(cn - tn);
            }
            phin = fn - mu * lg.wave_total(); thn = wave_sum(thn);
            int okf = (thn <= theta_max) && (phin == phin);
            {
                int bad = 0;
                for (int i = lane; i < nf; i += WAVE)
                    if (!(thn < c.s.Fth[i] || phin < c.s.Fph[i])) bad = 1;
                if (__any(bad)) okf = 0;
            }
            if (okf) {
                if (sw_try && al * sw_lhs > sw_rhs) {
                    if (phin <= phi0 + eta * al * Dphi + 10.0 * 2.2e-16 * fabs(phi0)) { acc = 1; ftype = 1; }
                } else if (thn <= (1.0 - 1e-5) * theta || phin <= phi0 - 1e-8 * theta) {
                    acc = 1;
                }
            }
            if (acc) break;
            al *= 0.5;
        }
        long long tc8 = clock64();
        tph[0] = tc1 - tc0; tph[1] = tc2 - tc1; tph[2] = tc3 - tc2; tph[3] = tc4 - tc3; tph[4] = tc5 - tc4; tph[5] = tc6 - tc5; tph[6] = tc7 - tc6; tph[7] = tc8 - tc7;
        if (kp.trace && b == kp.trace_problem && it < kp.trace_rows && lane == 0) {
            double* tr = kp.trace + (size_t)it * 16;
            for (int q = 0; q < 8; q++) tr[8 + q] = (double)(tph[q]);
            tr[0] = e_d; tr[1] = e_p; tr[2] = e_c; tr[3] = mu; tr[4] = al; tr[5] = a_d; tr[6] = dw; tr[7] = acc ? (ftype ? 2.0 : 1.0) : 0.0;
        }
        if (acc && !ftype && nf < MAXF) {
            if (lane == 0) { c.s.Fth[nf] = (1.0 - 1e-5) * theta; c.s.Fph[nf] = phi0 - 1e-8 * theta; }
            nf++;
        }
        if (!acc) break;
        // ---- accept ----------------------------------------------------------------------------------
        SYNC();
        for (int e = lane; e < (N + 1) * 6; e += WAVE) {
            int k = e / 6, i = e - k * 6;
            c.s.x[e] += al * c.s.dx[k * NX + i];
        }
        for (int e = lane; e < N * 2; e += WAVE) c.s.u[e] += al * c.s.du[(e >> 1) * NU + (e & 1)];
        if (NOBS)
            for (int e = lane; e < (N + 1) * NOBS; e += WAVE) {
                int k = e / NOBS, ob = e - k * NOBS;
                c.s.sg[e] += al * (k == 0 ? c.s.dx[6 + ob] : c.s.du[(k - 1) * NU + 2 + ob]);
            }
        for (int j = lane; j < m; j += WAVE) {
            if (!row_active(c, j)) continue;
            double tn = c.s.rtt[j];
            double nn = c.s.rnu[j] + a_d * c.s.rdnu[j];
            nn = fmin(fmax(nn, mu / (kappa_sigma * tn)), kappa_sigma * mu / tn);
            c.s.rt[j] = tn;
            c.s.rnu[j] = nn;
        }
        SYNC();
        // the step fields must not leak into the next evaluation at al = 0 (they are multiplied by al)
        f = fn;
        double numax = 0.0, th = 0.0;
        for (int j = lane; j < m; j += WAVE) {
            if (!row_active(c, j)) continue;
            double cv = c.s.rd[j] * row_value(c, j, 0.0);
            c.s.rc[j] = cv;
            numax = fmax(numax, c.s.rnu[j]);
            th = fmax(th, fabs(cv - c.s.rt[j]));
        }
        numax = wave_max(numax); th = wave_max(th);
        SYNC();
        assemble_first_order(c);
        if (numax > 1e12 && th > 1e-6) { status = 2; it++; break; }
    }
    if (infeas0) status = 2;

    // ---- write back: one coalesced pass ------------------------------------------------------------
    SYNC();
    double* Xb = kp.X + (size_t)b * (N + 1) * 6;
    double* Ub = kp.U + (size_t)b * N * 2;
    if (kp.mode == 0 && status != 0) {
        // reference fall-back trajectory (overtake_traj_planner.py:365-374)
        const double s0 = kp.x0[(size_t)b * 6 + 4], vx0 = kp.x0[(size_t)b * 6];
        double* bs = c.s.dx;
        double* be = c.s.dx + (N + 1);
        for (int j = lane; j <= N; j += WAVE) {
            bs[j] = kp.bez_s[(size_t)b * (N + 1) + j];
            be[j] = kp.bez_ey[(size_t)b * (N + 1) + j];
        }
        SYNC();
        for (int e = lane; e < (N + 1) * 6; e += WAVE) {
            int j = e / 6, i = e - j * 6;
            double st = s0 + kp.fallback_gain * j * kp.dt_ref * vx0;
            double v = 0.0;
            if (i == 0) v = kp.fallback_gain * vx0;
            else if (i == 4) v = st;
            else if (i == 5) v = interp_lin(bs, be, N + 1, fmin(fmax(st, bs[0]), bs[N]));
            Xb[e] = v;
        }
        for (int e = lane; e < N * 2; e += WAVE) Ub[e] = 0.0;
        if (lane == 0) kp.cost[b] = INFINITY;
    } else {
        for (int e = lane; e < (N + 1) * 6; e += WAVE) Xb[e] = c.s.x[e];
        for (int e = lane; e < N * 2; e += WAVE) Ub[e] = c.s.u[e];
        if (lane == 0) kp.cost[b] = f;
    }
    if (kp.mode == 1 && kp.sigma) {
        for (int e = lane; e < kp.n_obs_max * (N + 1); e += WAVE) {
            int ob = e / (N + 1), k = e - ob * (N + 1);
            kp.sigma[(size_t)b * kp.n_obs_max * (N + 1) + e] = (NOBS && ob < c.nobs) ? c.s.sg[k * no + ob] : 0.0;
        }
    }
    if (lane == 0) { kp.status[b] = status; kp.kkt[b] = E0; kp.iters[b] = it; }
}

// ------------------------------------------------------------------------------------------------
// region selection (planning/overtake_traj_planner.py:205-246): one wave per scenario, lanes over
// (region, side, stage) collision tests, ballot-free reductions per region.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) crx_select_kernel(const crx_select_kparams sp) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= sp.n_scen) return;
    const int N = sp.N, V = sp.V, R = V + 1, nv = sp.n_veh[s];
    const double r2 = sp.veh_length * sp.veh_length + sp.veh_width * sp.veh_width;
    double best_c = INFINITY;
    int best = 0;
    for (int r = 0; r < R; r++) {
        double cst = INFINITY;
        if (r <= nv) {
            const double* Xr = sp.X + (((size_t)s * R + r) * (N + 1)) * 6;
            double hits = 0.0;
            for (int e = lane; e < 2 * (N + 1); e += WAVE) {
                int side = e / (N + 1), j = e - side * (N + 1);
                int v = side == 0 ? r - 1 : r;                                              // :213, :227
                if (v < 0 || v >= nv) continue;
                double os = sp.obs_s[((size_t)s * V + v) * (N + 1) + j];
                while (os > sp.lap_length) os -= sp.lap_length;                             // :216-217
                double ds = Xr[6 * j + 4] - os;
                double de = Xr[6 * j + 5] - sp.obs_ey[((size_t)s * V + v) * (N + 1) + j];
                if (!(ds * ds + de * de - r2 >= 0.0)) hits += 1.0;                          // :220-223
            }
            hits = wave_sum(hits);
            cst = -sp.w_prog * (Xr[6 * N + 4] - Xr[4]) + sp.w_coll * hits;                  // :209
            if (sp.old_flag[s] >= 0 && sp.old_flag[s] != r) cst += sp.w_switch;             // :238-243
            if (cst < best_c) { best_c = cst; best = r; }                                   // first arg-min :244
        }
        if (lane == 0) sp.sel_cost[(size_t)s * R + r] = cst;
    }
    if (lane == 0) sp.flag[s] = best;
    const double* Xb = sp.X + (((size_t)s * R + best) * (N + 1)) * 6;
    for (int e = lane; e < (N + 1) * 6; e += WAVE) sp.best_X[(size_t)s * (N + 1) * 6 + e] = Xb[e];
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (plain C++ linkage inside the library; the C ABI lives in crx_api.hip)
// ------------------------------------------------------------------------------------------------
template <int NOBS>
static hipError_t launch_t(const crx_kparams& kp, hipStream_t st) {
    const int n = Lds<NOBS>::doubles(kp.N);
    size_t bytes = (size_t)n * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void*)crx_solve_kernel<NOBS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(crx_solve_kernel<NOBS>, dim3(kp.batch), dim3(WAVE), bytes, st, kp);
    return hipGetLastError();
}

hipError_t crx_launch_solve(const crx_kparams& kp, int nobs_template, hipStream_t st) {
    if (kp.batch == 0) return hipSuccess;
    switch (nobs_template) {
        case 0: return launch_t<0>(kp, st);
        case 1: return launch_t<1>(kp, st);
        case 2: return launch_t<2>(kp, st);
        case 3: return launch_t<3>(kp, st);
        default: return hipErrorInvalidValue;
    }
}

size_t crx_solve_lds_bytes(int N, int nobs_template) {
    switch (nobs_template) {
        case 0: return sizeof(double) * Lds<0>::doubles(N);
        case 1: return sizeof(double) * Lds<1>::doubles(N);
        case 2: return sizeof(double) * Lds<2>::doubles(N);
        default: return sizeof(double) * Lds<3>::doubles(N);
    }
}

hipError_t crx_launch_select(const crx_select_kparams& sp, hipStream_t st) {
    if (sp.n_scen == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_select_kernel, dim3(sp.n_scen), dim3(WAVE), 0, st, sp);
    return hipGetLastError();
}
