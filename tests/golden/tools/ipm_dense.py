"""Dense NumPy prototype of the interior-point method that crx implements (algorithm spec: DESIGN.md §4).

TEST TOOLING ONLY.  Two jobs:
  1. golden generation: solves the NLP exactly as recorded from the reference's own code by the
     CasADi shim (full-space variables, equality constraints eliminated by a dense null-space
     basis) -- a third, structurally different implementation next to oracle/ (condensed C) and
     the HIP kernel (Riccati);
  2. algorithm prototyping.
Every result is accepted only through nlp_solve.kkt_certificate, never on this solver's own flag.
"""
import numpy as np


class Opts:
    tol = 1e-8
    max_iter = 200
    mu0 = 0.1
    kappa_eps = 10.0
    kappa_mu = 0.2
    theta_mu = 1.5
    tau_min = 0.99
    slack_push = 1e-2
    nu0 = 1.0
    g_max = 100.0  # gradient-based row scaling target
    kappa_sigma = 1e10
    dw_first = 1e-4
    dw_min = 1e-20
    dw_max = 1e40
    kw_dec = 1.0 / 3.0
    kw_inc_first = 100.0
    kw_inc = 8.0
    eta = 1e-8
    rho0 = 1.0
    smax = 100.0
    verbose = False


def _chol_ok(H):
    try:
        L = np.linalg.cholesky(H)
        return L
    except np.linalg.LinAlgError:
        return None


def solve(fun, hess, v0, row_scale=None, opts=None):
    """minimise f(v) s.t. c(v) >= 0.

    fun(v) -> (f, g[n], c[m], J[m,n]);  hess(v, nu) -> W[n,n] = hess f - sum_j nu_j hess c_j
    row_scale: optional per-inequality scale d_j (c_j <- d_j c_j); multipliers are returned for the
    ORIGINAL rows.
    Returns dict(v, t, nu, status, iters, kkt, mu).
    status: 0 converged, 1 max_iter / line-search failure, 2 infeasible (diverging).
    """
    o = opts or Opts()
    v = np.array(v0, dtype=float)
    n = v.size
    f, g, c, J = fun(v)
    m = c.size
    d = np.ones(m) if row_scale is None else np.asarray(row_scale, dtype=float)
    c, J = d * c, d[:, None] * J
    t = np.maximum(c, o.slack_push)
    nu = np.full(m, o.nu0)
    mu = o.mu0
    dw_last = 0.0
    rho = o.rho0
    status = 1
    it = 0
    hist = []
    best_theta = np.inf
    for it in range(o.max_iter + 1):
        rd = g - J.T @ nu
        rp = c - t
        sd = max(o.smax, np.abs(nu).sum() / max(m, 1)) / o.smax
        e_d = np.abs(rd).max() / sd if n else 0.0
        e_p = np.abs(rp).max() if m else 0.0
        e_c0 = np.abs(t * nu).max() / sd if m else 0.0
        E0 = max(e_d, e_p, e_c0)
        hist.append((it, f, E0, mu))
        if o.verbose:
            print("it %3d f %.8e  ed %.2e ep %.2e ec %.2e mu %.1e dw %.1e" % (it, f, e_d, e_p, e_c0, mu, dw_last))
        if E0 <= o.tol:
            status = 0
            break
        if it == o.max_iter:
            break
        # barrier-parameter update (monotone, Fiacco-McCormick)
        while True:
            e_cm = np.abs(t * nu - mu).max() / sd if m else 0.0
            Emu = max(e_d, e_p, e_cm)
            if Emu <= o.kappa_eps * mu and mu > o.tol / 10.0:
                mu = max(o.tol / 10.0, min(o.kappa_mu * mu, mu ** o.theta_mu))
                rho = o.rho0
            else:
                break
        tau = max(o.tau_min, 1.0 - mu)
        # Newton system in the primal variables
        Sig = nu / t
        W = hess(v, nu * d)
        Hb = W + J.T @ (Sig[:, None] * J)
        rhs = -(g - J.T @ (mu / t - Sig * rp))
        dw = 0.0
        L = _chol_ok(Hb)
        if L is None:
            dw = o.dw_first if dw_last == 0.0 else max(o.dw_min, o.kw_dec * dw_last)
            while True:
                L = _chol_ok(Hb + dw * np.eye(n))
                if L is not None:
                    break
                dw *= o.kw_inc_first if dw_last == 0.0 else o.kw_inc
                if dw > o.dw_max:
                    break
            if L is None:
                status = 1
                break
            dw_last = dw
        dv = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        dt = J @ dv + rp
        dnu = (mu - t * nu - nu * dt) / t
        # fraction to the boundary
        neg = dt < 0
        a_p = min(1.0, (-tau * t[neg] / dt[neg]).min()) if neg.any() else 1.0
        neg = dnu < 0
        a_d = min(1.0, (-tau * nu[neg] / dnu[neg]).min()) if neg.any() else 1.0
        # l1 merit line search
        theta = np.abs(rp).sum()
        Dphi = g @ dv - mu * (dt / t).sum()
        curv = dv @ (Hb @ dv) + dw * (dv @ dv)
        if theta > 0:
            rho_trial = (Dphi + 0.5 * max(curv, 0.0)) / (0.9 * theta)
            if rho < rho_trial:
                rho = rho_trial + 1.0
        DM = Dphi - rho * theta
        M0 = f - mu * np.log(t).sum() + rho * theta
        a = a_p
        ok = False
        for _ in range(40):
            vn = v + a * dv
            tn = t + a * dt
            fn, gn, cn, Jn = fun(vn)
            cn, Jn = d * cn, d[:, None] * Jn
            tn = np.maximum(tn, cn)  # slack reset: never hurts theta nor the barrier
            Mn = fn - mu * np.log(tn).sum() + rho * np.abs(cn - tn).sum()
            if Mn <= M0 + o.eta * a * DM + 1e-13 * abs(M0):
                ok = True
                break
            a *= 0.5
        if not ok:
            status = 1
            break
        v, t, f, g, c, J = vn, tn, fn, gn, cn, Jn
        nu = nu + a_d * dnu
        nu = np.minimum(np.maximum(nu, mu / (o.kappa_sigma * t)), o.kappa_sigma * mu / t)
        # divergence test for infeasible problems (only meaningful when c is affine)
        th = np.abs(c - t).max() if m else 0.0
        best_theta = min(best_theta, th)
        if np.abs(nu).max() > 1e12 and th > 1e-6:
            status = 2
            break
    return dict(v=v, t=t, nu=nu * d, status=status, iters=it, kkt=E0, mu=mu, f=f, hist=hist)


# -------------------------------------------------------------------------------------------------
# glue for problems recorded by the CasADi shim
# -------------------------------------------------------------------------------------------------
_KEY = [10 ** 9]


def _jac_rows(opti, nodes, rows, z):
    _KEY[0] += 1
    out = np.zeros((len(rows), opti.nvar))
    for k, j in enumerate(rows):
        g = nodes[j].ev(z, _KEY[0])[1]
        if not np.isscalar(g):
            out[k] = g
    return out


def solve_recorded(opti, opts=None, v_start=None):
    n = opti.nvar
    f0, g0, ce0, Je, ci0, Ji0 = opti.eval_all(np.zeros(n))
    # particular solution + null-space basis of the (affine) equality constraints
    U, S, Vt = np.linalg.svd(Je)
    r = int((S > 1e-10 * S[0]).sum())
    assert r == Je.shape[0], "rank-deficient equality Jacobian"
    Z = Vt[r:].T
    zp = np.linalg.lstsq(Je, -ce0, rcond=None)[0]
    # constant rows (constraints on fixed quantities) are checked once and removed
    a = opti.eval_all(np.random.default_rng(1).normal(size=n))
    b = opti.eval_all(np.random.default_rng(2).normal(size=n) * 2.0)
    JZ_a = a[5] @ Z
    JZ_b = b[5] @ Z
    live = np.where((np.abs(JZ_a).max(axis=1) > 1e-13) | (np.abs(JZ_b).max(axis=1) > 1e-13))[0]
    dead = np.setdiff1d(np.arange(len(ci0)), live)
    c_at_p = opti.eval_all(zp)[4]
    const_violation = float(max(0.0, -c_at_p[dead].min())) if len(dead) else 0.0
    nl = np.where(np.abs(a[5] - b[5]).max(axis=1) > 1e-12)[0]
    nl_live = np.array([np.where(live == j)[0][0] for j in nl if j in live], dtype=int)
    touched = np.where(np.abs(a[5][nl]).max(axis=0) + np.abs(b[5][nl]).max(axis=0) > 0)[0] if len(nl) else np.array([], int)
    ineq_nodes = [c.node for c in opti.cons if c.kind == "ge"]
    # constant objective Hessian
    Hf = np.zeros((n, n))
    for i in range(n):
        e = np.zeros(n)
        e[i] = 1.0
        Hf[:, i] = opti.eval_all(e)[1] - g0
    Hf = 0.5 * (Hf + Hf.T)
    Hf_red = Z.T @ Hf @ Z

    def fun(v):
        z = zp + Z @ v
        f, g, ce, _, ci, Ji = opti.eval_all(z)
        return f, Z.T @ g, ci[live], Ji[live] @ Z

    def hess(v, nu):
        if len(nl_live) == 0:
            return Hf_red
        z = zp + Z @ v
        # exact-to-1e-9 Hessian of sum_j nu_j c_j over the nonlinear rows by central differences
        # of the AD Jacobian, only in the coordinates those rows touch
        Hc = np.zeros((n, n))
        w = np.zeros(len(ci0))
        w[live[nl_live]] = nu[nl_live]
        h = 1e-6
        wn = w[nl]
        for i in touched:
            e = np.zeros(n)
            e[i] = h * max(1.0, abs(z[i]))
            Jp = _jac_rows(opti, ineq_nodes, nl, z + e)
            Jm = _jac_rows(opti, ineq_nodes, nl, z - e)
            Hc[:, i] = (wn @ Jp - wn @ Jm) / (2 * e[i])
        Hc = 0.5 * (Hc + Hc.T)
        return Hf_red - Z.T @ Hc @ Z

    # gradient-based row scaling measured in the reference's (full-space) variables at the start
    if v_start is None:
        v_start = np.zeros(Z.shape[1])
    z_s = zp + Z @ v_start
    Js = opti.eval_all(z_s)[5][live]
    o = opts or Opts()
    d = np.minimum(1.0, o.g_max / np.maximum(np.abs(Js).max(axis=1), 1e-300))
    res = solve(fun, hess, v_start, row_scale=d, opts=o)
    z = zp + Z @ res["v"]
    nu_full = np.zeros(len(ci0))
    nu_full[live] = res["nu"]
    res.update(z=z, nu_full=nu_full, const_violation=const_violation, n_red=Z.shape[1], n_live=len(live))
    return res
