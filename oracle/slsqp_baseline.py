"""B2 of SURVEY.md section 8d: scipy.optimize SLSQP on a subsample of the MPC-CBF NLPs, as a sanity floor for the CPU
baseline (the only general NLP solver this image has; CasADi/IPOPT, the reference's own, is not installable).

TEST INFRASTRUCTURE like everything under oracle/: bench.py's cpu_baseline leg and tests only.

The NLP is control.mpccbf's (control/control.py:492-591) in condensed form: variables u (2N) and the CBF slacks
sigma (n_obs (N+1)); the states follow from x_{k+1} = A x_k + B u_k.  Zero start (the reference sets no initial guess).
Analytic gradients; SLSQP's own exit flag is not trusted -- the caller compares costs."""
import time

import numpy as np
from scipy.optimize import minimize


def _condense(A, B, N):
    Phi = np.zeros((N + 1, 6, 6))
    Gam = np.zeros((N + 1, 6, 2 * N))
    Phi[0] = np.eye(6)
    for k in range(N):
        Phi[k + 1] = A @ Phi[k]
        Gam[k + 1] = A @ Gam[k]
        Gam[k + 1][:, 2 * k:2 * k + 2] += B
    return Phi, Gam


def solve_one(desc, x0, xt, obs_s, obs_ey, lap_off, n_obs):
    N, q = int(desc.N), int(desc.degree)
    A = np.array(desc.A).reshape(6, 6)
    B = np.array(desc.B).reshape(6, 2)
    Q, R = np.array(desc.Q), np.array(desc.R)
    al, cm, Ls, Ws, wsig = desc.alpha, 1.0 + desc.margin, desc.l_sum, desc.w_sum, desc.w_slack
    Phi, Gam = _condense(A, B, N)
    xfree = np.einsum("kij,j->ki", Phi, x0)
    nu, ns = 2 * N, n_obs * (N + 1)
    xt = np.broadcast_to(xt, (N + 1, 6))

    def states(z):
        return xfree + Gam @ z[:nu]

    def f(z):
        X = states(z)
        u = z[:nu].reshape(N, 2)
        return float((Q * (X - xt) ** 2).sum() + (R * u ** 2).sum() + wsig * z[nu:].sum())

    def fg(z):
        X = states(z)
        g = np.zeros_like(z)
        g[:nu] = 2.0 * np.einsum("kia,ki->a", Gam, Q * (X - xt)) + 2.0 * (np.tile(R, N) * z[:nu])
        g[nu:] = wsig
        return g

    def cons(z):
        X = states(z)
        sig = z[nu:].reshape(n_obs, N + 1)
        out = []
        for o in range(n_obs):
            dsc = (X[:-1, 4] - obs_s[o, :-1] - lap_off[o]) / Ls          # :539-540
            dec = (X[:-1, 5] - obs_ey[o, :-1]) / Ws
            dsn = (X[1:, 4] - obs_s[o, 1:]) / Ls                          # :542 (quirk Q1)
            den = (X[1:, 5] - obs_ey[o, 1:]) / Ws
            out.append(dsn ** q + den ** q - sig[o, 1:] - (1 - al) * (dsc ** q + dec ** q - sig[o, :-1]) - al * cm)
        out.append(X[:, 0] - desc.v_min); out.append(desc.v_max - X[:, 0])   # :582-586 (k = 0 rows are constants)
        out.append(X[:, 5] + desc.ey_max); out.append(desc.ey_max - X[:, 5])
        return np.concatenate(out)

    def cons_jac(z):
        X = states(z)
        rows = []
        for o in range(n_obs):
            dsc = (X[:-1, 4] - obs_s[o, :-1] - lap_off[o]) / Ls
            dec = (X[:-1, 5] - obs_ey[o, :-1]) / Ws
            dsn = (X[1:, 4] - obs_s[o, 1:]) / Ls
            den = (X[1:, 5] - obs_ey[o, 1:]) / Ws
            J = np.zeros((N, nu + ns))
            J[:, :nu] = (q * dsn ** (q - 1) / Ls)[:, None] * Gam[1:, 4] + (q * den ** (q - 1) / Ws)[:, None] * Gam[1:, 5] \
                - (1 - al) * ((q * dsc ** (q - 1) / Ls)[:, None] * Gam[:-1, 4] + (q * dec ** (q - 1) / Ws)[:, None] * Gam[:-1, 5])
            for i in range(N):
                J[i, nu + o * (N + 1) + i + 1] = -1.0
                J[i, nu + o * (N + 1) + i] = 1 - al
            rows.append(J)
        for comp, sgn in ((0, 1.0), (0, -1.0), (5, 1.0), (5, -1.0)):
            J = np.zeros((N + 1, nu + ns))
            J[:, :nu] = sgn * Gam[:, comp]
            rows.append(J)
        return np.vstack(rows)

    lo = np.concatenate([np.tile([-desc.delta_max, -desc.a_max], N), np.zeros(ns)])
    hi = np.concatenate([np.tile([desc.delta_max, desc.a_max], N), np.full(ns, np.inf)])
    r = minimize(f, np.zeros(nu + ns), jac=fg, bounds=list(zip(lo, hi)), method="SLSQP",
                 constraints=[dict(type="ineq", fun=cons, jac=cons_jac)], options=dict(maxiter=300, ftol=1e-12))
    return r.x, float(r.fun), float(min(cons(r.x).min(), 0.0))


def time_batch(desc, p, n=64):
    """Solve the first n problems of a crx.synth cfg2-type batch one after the other; returns (solves/s, costs, viol)."""
    t0 = time.perf_counter()
    costs, viol = [], []
    for b in range(n):
        nb = int(p["n_obs"][b])
        _, c, v = solve_one(desc, p["x0"][b], p["xt"][b], p["obs_s"][b, :nb], p["obs_ey"][b, :nb], p["lap_off"][b, :nb], nb)
        costs.append(c); viol.append(v)
    el = time.perf_counter() - t0
    return n / el, np.array(costs), np.array(viol)
