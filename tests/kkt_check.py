"""Oracle-free first-order check of a point of the MPC-CBF NLP (numpy + scipy only; nothing from oracle/ or libcrx).

The NLP is the reference's (car_racing/control/control.py:492-591 `mpccbf`, :270-382 `mpc_multi_agents`), written in the reduced
variables z = (U, sigma): the states follow from x_0 and the LTI dynamics (:566-570).  Rows, all as c(z) >= 0:
  input box (:572-576), vx / ey box at stages 1..N (:582-586; the rows on the fixed x_0 are constants), sigma >= 0 (:559,561),
  CBF rows h_{i+1} - (1 - alpha) h_i >= 0 with h_i = (ds_i / L)^q + (dey_i / W)^q - 1 - margin - sigma_i (:527-558; ds_i of the
  "current" term is lap-corrected, of the "next" term not: quirk Q1).
`certificate` answers "is this a KKT point at IPOPT's own tolerances?" without trusting any solver's multipliers: it looks for
nu >= 0 with nu_j c_j <= compl_tol (IPOPT compl_inf_tol, 1e-4) minimising || grad f - J' nu ||_inf, a linear program over
ALL rows (no active-set threshold: an interior-point answer leaves rows with a slack of 1e-6 and a multiplier of 1 -- inside the
complementarity tolerance, outside any fixed activity threshold; DESIGN.md section 3, problem #99).
Used by tests/test_gpu_parity.py to CLASSIFY pairs of solves (kernel vs oracle) that ended at different points: two different certified
points of a non-convex NLP are two local solutions, not a parity failure.
"""
import numpy as np


def _desc_fields(d):
    A = np.array(d.A, dtype=float).reshape(6, 6)
    B = np.array(d.B, dtype=float).reshape(6, 2)
    return dict(N=int(d.N), V=int(d.n_obs_max), A=A, B=B, Q=np.array(d.Q, dtype=float), R=np.array(d.R, dtype=float), q=int(d.degree),
                umax=np.array([d.delta_max, d.a_max]), vmin=d.v_min, vmax=d.v_max, eymax=d.ey_max, alpha=d.alpha, cm=1.0 + d.margin,
                L=d.l_sum, W=d.w_sum, ws=d.w_slack, pst=bool(d.per_stage_target))


def problem(d, x0, xt, obs_s, obs_ey, lap_off, n_obs):
    """One problem (rows of the batch arrays) -> closures over z = [U (N x 2) row-major, sigma (n_obs x (N+1)) row-major]."""
    P = _desc_fields(d)
    N, A, B, n = P["N"], P["A"], P["B"], int(n_obs)
    nu_ = 2 * N
    nz = nu_ + n * (N + 1)
    # x_k = Phi_k x0 + sum_j G[k][:, 2j:2j+2] u_j
    Ap = [np.eye(6)]
    for _ in range(N):
        Ap.append(A @ Ap[-1])
    G = np.zeros((N + 1, 6, nu_))
    for k in range(1, N + 1):
        for j in range(k):
            G[k][:, 2 * j:2 * j + 2] = Ap[k - 1 - j] @ B
    free = np.array([Ap[k] @ np.asarray(x0, dtype=float) for k in range(N + 1)])
    xt = np.asarray(xt, dtype=float)
    xref = xt.reshape(N + 1, 6) if P["pst"] else np.tile(xt.reshape(6), (N + 1, 1))
    q, L, W, cm, om = P["q"], P["L"], P["W"], P["cm"], 1.0 - P["alpha"]

    def split(z):
        return z[:nu_].reshape(N, 2), z[nu_:].reshape(n, N + 1)

    def states(z):
        return free + np.einsum("kij,j->ki", G, z[:nu_])

    def cost(z):
        U, S = split(z)
        X = states(z)
        return float((U * U * P["R"]).sum() + (((X - xref) ** 2) * P["Q"]).sum() + P["ws"] * S.sum())

    def grad(z):
        U, S = split(z)
        X = states(z)
        g = np.zeros(nz)
        g[:nu_] = (2.0 * U * P["R"]).reshape(-1) + np.einsum("kij,ki->j", G, 2.0 * (X - xref) * P["Q"])
        g[nu_:] = P["ws"]
        return g

    def rows(z):
        """c (m,), J (m, nz), kinds (m,) -- every row as c >= 0."""
        U, S = split(z)
        X = states(z)
        c, J, kind = [], [], []
        for k in range(N):
            for i in range(2):
                e = np.zeros(nz); e[2 * k + i] = 1.0
                c += [U[k, i] + P["umax"][i], P["umax"][i] - U[k, i]]; J += [e, -e]; kind += ["u", "u"]
        for k in range(1, N + 1):
            for comp, lo, hi in ((0, P["vmin"], P["vmax"]), (5, -P["eymax"], P["eymax"])):
                r = np.zeros(nz); r[:nu_] = G[k][comp]
                c += [X[k, comp] - lo, hi - X[k, comp]]; J += [r, -r]; kind += ["x", "x"]
        for o in range(n):
            for k in range(N + 1):
                e = np.zeros(nz); e[nu_ + o * (N + 1) + k] = 1.0
                c.append(S[o, k]); J.append(e); kind.append("s")
        for o in range(n):
            de = (X[:, 5] - obs_ey[o]) / W
            dsn = (X[:, 4] - obs_s[o]) / L
            dsc = (X[:, 4] - obs_s[o] - lap_off[o]) / L
            for i in range(N):
                hn = dsn[i + 1] ** q + de[i + 1] ** q - cm - S[o, i + 1]
                hc = dsc[i] ** q + de[i] ** q - cm - S[o, i]
                c.append(hn - om * hc)
                r = np.zeros(nz)
                r[:nu_] = (q * dsn[i + 1] ** (q - 1) / L) * G[i + 1][4] + (q * de[i + 1] ** (q - 1) / W) * G[i + 1][5] \
                    - om * ((q * dsc[i] ** (q - 1) / L) * G[i][4] + (q * de[i] ** (q - 1) / W) * G[i][5])
                r[nu_ + o * (N + 1) + i + 1] = -1.0
                r[nu_ + o * (N + 1) + i] = om
                J.append(r); kind.append("cbf")
        return np.array(c), np.array(J), np.array(kind)

    def pack(U, sigma):
        return np.concatenate([np.asarray(U, dtype=float).reshape(-1), np.asarray(sigma, dtype=float)[:n].reshape(-1)])

    return dict(cost=cost, grad=grad, rows=rows, pack=pack, states=states, nz=nz, n_obs=n)


def certificate(prob, z, compl_tol=1e-4):
    """Best multipliers nu >= 0 with nu_j c_j <= compl_tol (rows at or beyond their bound: unbounded above) for the point z: the LINEAR PROGRAM
    min t  s.t.  -t <= grad f - J' nu <= t,  0 <= nu_j <= compl_tol / c_j  (HiGHS; rows equilibrated by their largest gradient entry: a CBF row's is
    1e2..1e9).  Returns dict(stationarity = ||grad f - J' nu||_inf, grad_scale = max(1, ||grad f||_inf), violation = max(0, -min c) on the
    equilibrated rows, complementarity, nu_max, cost)."""
    from scipy.optimize import linprog
    g = prob["grad"](z)
    c, J, kind = prob["rows"](z)
    s = np.maximum(1.0, np.abs(J).max(axis=1))       # nu = nu_s / s, J_s = J / s, c_s = c / s: nu_s c_s = nu c
    Js, cs = J / s[:, None], c / s
    m, n = Js.shape
    gs = max(1.0, float(np.abs(g).max()))
    # variables [nu_s / gs (m), t / gs]: the LP is solved on the gradient's scale
    ub = [(0.0, None if cs[j] <= 0.0 else compl_tol / cs[j] / gs) for j in range(m)] + [(0.0, None)]
    A_ub = np.block([[-Js.T, -np.ones((n, 1))], [Js.T, -np.ones((n, 1))]])
    b_ub = np.concatenate([-g / gs, g / gs])
    res = linprog(np.concatenate([np.zeros(m), [1.0]]), A_ub=A_ub, b_ub=b_ub, bounds=ub, method="highs")
    if res.status != 0:
        return dict(stationarity=float("inf"), grad_scale=gs, violation=float(max(0.0, -cs.min())), complementarity=float("nan"), nu_max=float("nan"),
                    cost=prob["cost"](z), lp_status=int(res.status))
    nus = res.x[:m] * gs
    r = g - Js.T @ nus
    return dict(stationarity=float(np.abs(r).max()), grad_scale=gs, violation=float(max(0.0, -cs.min())),
                complementarity=float((nus * np.maximum(cs, 0.0)).max()), nu_max=float((nus / s).max()), cost=prob["cost"](z), lp_status=0)


def is_kkt_point(cert, rel=1e-6, viol=1e-6):
    """A KKT point at IPOPT's tolerances: rows hold to `viol` on the equilibrated rows, and the best admissible multipliers (complementarity within
    IPOPT's compl_inf_tol) leave a stationarity residual below `rel` of the cost gradient's size (1e4: the slack weight) -- IPOPT's own unscaled
    bound is dual_inf_tol = 1, i.e. 1e-4 of it."""
    return cert["violation"] <= viol and cert["stationarity"] <= rel * cert["grad_scale"]


KEYS = ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")


def classify_pairs(d, p, ra, rb, names=("gpu", "oracle"), dx_tol=1e-5):
    """Two solves of the same batch (result dicts with X, U, sigma, cost, status).  Every problem both sides report CONVERGED on whose trajectories differ by
    more than dx_tol goes through the certificate on both points.  Returns (lines, table): table counts 'both_kkt' (two local solutions of a non-convex
    NLP; which side is cheaper is recorded), 'same_cost' (both certified, costs equal to 1e-6: a flat direction -- the uncosted states vy / wz of a crash
    state), 'uncertified' (at least one point fails: a parity failure)."""
    both = (ra["status"] == 0) & (rb["status"] == 0)
    dx = np.abs(ra["X"] - rb["X"]).reshape(len(both), -1).max(axis=1)
    idx = np.nonzero(both & (dx > dx_tol))[0]
    table = dict(pairs=int(len(idx)), both_kkt=0, same_cost=0, uncertified=0, a_cheaper=0, b_cheaper=0, cost_ratio_max=1.0)
    lines = []
    for b in idx:
        pr = problem(d, *[p[k][b] for k in KEYS])
        ca, cb = (certificate(pr, pr["pack"](r["U"][b], r["sigma"][b])) for r in (ra, rb))
        ok = is_kkt_point(ca) and is_kkt_point(cb)
        rel = abs(ca["cost"] - cb["cost"]) / max(1.0, abs(ca["cost"]), abs(cb["cost"]))
        ratio = max(ca["cost"], cb["cost"]) / max(min(ca["cost"], cb["cost"]), 1e-300)
        if not ok:
            table["uncertified"] += 1
        elif rel <= 1e-6:
            table["same_cost"] += 1
        else:
            table["both_kkt"] += 1
            table["a_cheaper" if ca["cost"] < cb["cost"] else "b_cheaper"] += 1
            table["cost_ratio_max"] = max(table["cost_ratio_max"], ratio)
        lines.append("#%d |dX| %.1e  cost %s %.9g / %s %.9g (ratio %.3g)  stationarity / grad %.1e | %.1e  rows %.1e | %.1e  -> %s" % (
            b, dx[b], names[0], ca["cost"], names[1], cb["cost"], ratio, ca["stationarity"] / ca["grad_scale"], cb["stationarity"] / cb["grad_scale"],
            ca["violation"], cb["violation"], "two KKT points" if ok and rel > 1e-6 else ("same cost, flat direction" if ok else "NOT CERTIFIED")))
    return lines, table
