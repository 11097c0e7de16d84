"""The CPU oracle (oracle/crx_oracle.c) against the fixtures recorded from the reference's own code
(tests/golden/tools/make_golden.py).  This is the pin that makes the oracle trustworthy; the GPU
parity tests then compare libcrx with the oracle.

The goldens are certified KKT points solved to 1e-11.  At the product's default tol (1e-8, IPOPT's)
the interior-point perturbation mu ~ 1e-9 moves the weakly determined states (vy, wz carry no cost)
by up to ~1e-5 and the cost by ~1e-8 relative; solved to 1e-11 everything agrees to ~1e-7.
"""
import numpy as np
import pytest

import helpers

TIGHT = dict(tol=1e-11, x=2e-7, u=2e-6, f=1e-9, xw=1e-7)
DEFAULT = dict(tol=1e-8, x=5e-4, u=2e-3, f=1e-7, xw=1e-5)  # xw: the cost-weighted states vx, s, ey
TOL_FALLBACK = 1e-12  # closed-form fall-back trajectories


def _with_tol(d, tol):
    d.opts.tol = tol
    return d


def _check_X(X, Xg, T, tag):
    np.testing.assert_allclose(X, Xg, atol=T["x"], err_msg=tag)
    np.testing.assert_allclose(X[:, [0, 4, 5]], Xg[:, [0, 4, 5]], atol=T["xw"], err_msg=tag)


def _close_cost(a, b, rel):
    return abs(a - b) <= rel * max(1.0, abs(b))


@pytest.mark.parametrize("T", [DEFAULT, TIGHT], ids=["tol1e-8", "tol1e-11"])
def test_mpccbf_cases(orc, AB, golden_mpccbf, T):
    A, B = AB
    for name in golden_mpccbf.names:
        g = golden_mpccbf.case(name)
        d, args = helpers.mpccbf_inputs(g, A, B)
        r = orc.cbf_solve(_with_tol(d, T["tol"]), *args)
        # the window filter must keep exactly the obstacles the reference put into its NLP
        assert int(args[-1][0]) == int(g["n_obs_in_problem"]), name
        if not bool(g["success"]):
            assert r["status"][0] != 0, name
            continue
        assert r["status"][0] == 0, (name, r["status"], r["kkt"], r["iters"])
        assert r["kkt"][0] <= T["tol"]
        assert _close_cost(r["cost"][0], g["cert"][0], T["f"]), (name, r["cost"][0], g["cert"][0])
        _check_X(r["X"][0], g["X"], T, name)
        np.testing.assert_allclose(r["U"][0], g["U"], atol=T["u"], err_msg=name)
        # what the reference's mpccbf handed back to its caller: u_pred[0,:] (control.py:607)
        np.testing.assert_allclose(r["U"][0, 0], g["u_returned"], atol=T["u"], err_msg=name)
        n = int(g["n_obs_in_problem"])
        if n:
            np.testing.assert_allclose(r["sigma"][0, :n], g["sigma"], atol=1e-6, err_msg=name)


@pytest.mark.parametrize("T", [DEFAULT, TIGHT], ids=["tol1e-8", "tol1e-11"])
def test_planner_regions(orc, AB, golden_planner, T):
    A, B = AB
    seen_fail = seen_ok = 0
    for name in golden_planner.names:
        g = golden_planner.case(name)
        if not bool(g["overtake_flag"]):
            continue
        d, args = helpers.planner_inputs(g, A, B)
        r = orc.planner_solve(_with_tol(d, T["tol"]), *args)
        for reg, ok in enumerate(g["region_success"]):
            tag = "%s/region%d" % (name, reg)
            if ok:
                seen_ok += 1
                assert r["status"][reg] == 0, (tag, r["status"], r["kkt"], r["iters"])
                assert _close_cost(r["cost"][reg], g["region_cert"][reg, 0], T["f"]), tag
                _check_X(r["X"][reg], g["region_X"][reg], T, tag)
            else:
                # HiGHS proved the recorded QP infeasible -> the reference takes :365-374
                seen_fail += 1
                assert r["status"][reg] != 0, tag
                assert np.isinf(r["cost"][reg])
                np.testing.assert_allclose(r["X"][reg], g["region_X"][reg], atol=TOL_FALLBACK, err_msg=tag)
    assert seen_ok >= 10 and seen_fail >= 2


def test_selection(orc, golden_planner):
    from crx import abi

    for name in golden_planner.names:
        g = golden_planner.case(name)
        if not bool(g["overtake_flag"]):
            continue
        N = int(g["N"])
        V = g["obs_pred"].shape[0]
        d = abi.select_desc(N, V, float(g["lap_length"]))
        r = orc.select(d, np.array([V]), g["region_X"][None], g["obs_pred"][None, :, 4, :],
                       g["obs_pred"][None, :, 5, :], np.array([int(g["old_flag"])]))
        assert int(r["flag"][0]) == int(g["direction_flag"]), name
        np.testing.assert_allclose(r["best_X"][0], g["traj_xcurv"], atol=1e-12)


@pytest.mark.parametrize("T", [DEFAULT, TIGHT], ids=["tol1e-8", "tol1e-11"])
def test_mpc_multi_agents(orc, AB, golden_planner, T):
    A, B = AB
    n = 0
    for name in golden_planner.names:
        g = golden_planner.case(name)
        if not bool(g["overtake_flag"]) or not bool(g["mma_present"]):
            continue
        d, args = helpers.mma_inputs(g, A, B)
        r = orc.cbf_solve(_with_tol(d, T["tol"]), *args)
        assert int(args[-1][0]) == int(g["mma_n_obs"]), name
        assert bool(g["mma_success"])
        assert r["status"][0] == 0, (name, r["status"], r["kkt"], r["iters"])
        assert _close_cost(r["cost"][0], g["mma_cert"][0], T["f"]), (name, r["cost"][0], g["mma_cert"][0])
        _check_X(r["X"][0], g["mma_X"], T, name)
        np.testing.assert_allclose(r["U"][0, 0], g["mma_u"], atol=T["u"], err_msg=name)
        # what mpc_multi_agents returned: (u_pred[0,:], x_pred) (control.py:473)
        np.testing.assert_allclose(r["X"][0], g["mma_x_pred"], atol=T["x"], err_msg=name)
        n += 1
    assert n >= 7


@pytest.mark.parametrize("linalg", [0, 1], ids=["kkt-lu", "block-cholesky"])
def test_lmpc_qps(orc, golden_racing_game, linalg):
    """Learning-MPC QPs (control.py:610-730) exactly as the reference built them in its LMPC lap --
    LTV model from its own safe-set regression, safe-set hull, Q-function -- against the certified
    solutions.  Both linear-algebra routes of the oracle (full KKT LU; the kernel's block Cholesky)."""
    g = golden_racing_game
    d, args = helpers.lmpc_inputs(g)
    orc.lib.crx_oracle_lmpc_set_linalg(linalg)
    try:
        r = orc.lmpc_solve(d, *args)
    finally:
        orc.lib.crx_oracle_lmpc_set_linalg(0)
    ok = g["lmpc_success"]
    assert ok.sum() >= 30
    assert (r["status"][ok] == 0).all(), r["status"][ok]
    assert r["kkt"][ok].max() <= 1e-8
    # feasible instances: strictly convex in u -> unique (x, u)
    assert np.abs(r["X"][ok] - g["lmpc/X"][ok]).max() <= 5e-6
    assert np.abs(r["U"][ok] - g["lmpc/U"][ok]).max() <= 5e-6
    rel = np.abs(r["cost"][ok] - g["lmpc/cert"][ok, 0]) / np.abs(g["lmpc/cert"][ok, 0])
    assert rel.max() <= 1e-8
    # the terminal state sits in the hull of the selected safe-set points: x_N = SS lambd, lambd in the simplex
    lam = r["lam"][ok]
    assert lam.min() >= -1e-9 and np.abs(lam.sum(1) - 1).max() <= 1e-8
    xN = np.einsum("bcm,bm->bc", g["lmpc/ss"][ok], lam)
    assert np.abs(xN - r["X"][ok][:, -1]).max() <= 1e-7
    # the first instance HiGHS proved infeasible (the reference pins its terminal slack to zero): reported, not hidden
    first_bad = int(g["lmpc_first_uncertified"])
    assert r["status"][first_bad] == 2


def test_plant_step(orc):
    """The plant restatement (system/vehicle_dynamics.py:4-49 under the sub-step loop of
    utils/base.py:897-942) against the reference: 40 recorded single Euler steps and the 30-step PID
    closed loop on l_shape (tests/golden/harness.npz)."""
    import os

    import conftest
    from control import control
    from crx import abi
    from utils import racing_env

    H = np.load(os.path.join(conftest.GOLDEN, "harness.npz"))
    for i in range(H["plant/u"].shape[0]):
        d = abi.plant_desc(1, 1e9, timestep=0.001)          # one sub-step, one segment carrying the recorded curvature
        tr = np.array([[0, 0, 0, -1e9, 3e9, H["plant/curv"][i]]])
        r = orc.plant_step(d, tr, H["plant/xglob"][i:i + 1], H["plant/xcurv"][i:i + 1], H["plant/u"][i:i + 1])
        np.testing.assert_allclose(r["xglob"][0], H["plant/xglob_next"][i], rtol=0, atol=1e-15)
        np.testing.assert_allclose(r["xcurv"][0], H["plant/xcurv_next"][i], rtol=0, atol=1e-15)
    spec = np.genfromtxt(os.path.join(conftest.ROOT, "data/track_layout/l_shape.csv"), delimiter=",")
    track = racing_env.ClosedTrack(spec, track_width=0.8)
    d = abi.plant_desc(track.point_and_tangent.shape[0], track.lap_length)
    assert d.n_sub == 100
    xg, xc = np.zeros((1, 6)), np.zeros((1, 6))
    for k in range(H["pid/xcurv_log"].shape[0]):
        u = control.pid(xc[0], np.array([0.8, 0, 0, 0, 0, 0.0]))
        r = orc.plant_step(d, track.point_and_tangent, xg, xc, u[None])
        xg, xc = r["xglob"], r["xcurv"]
        np.testing.assert_allclose(xc[0], H["pid/xcurv_log"][k], rtol=0, atol=1e-13)
        np.testing.assert_allclose(xg[0], H["pid/xglob_log"][k], rtol=0, atol=1e-13)


def test_path_planner_regions(orc, golden_path):
    """OvertakePathPlanner QPs (overtake_path_planner.py:199-318) as the reference built them (recorded through the
    CasADi stand-in, certified): every region's verdict and solution, and the selection the reference made."""
    seen_ok = seen_bad = 0
    for name in golden_path.names:
        c = golden_path.case(name)
        if not bool(c["overtake_flag"]):
            continue
        d, qp = helpers.path_inputs(c)
        r = orc.path_solve(d, *qp)
        ok = c["region_success"]
        assert ((r["status"] == 0) == ok).all(), (name, r["status"], ok)
        np.testing.assert_allclose(r["E"][ok], c["region_E"][ok], rtol=0, atol=2e-6, err_msg=name)
        np.testing.assert_allclose(r["cost"][ok], c["region_cert"][ok, 0], rtol=1e-8, err_msg=name)
        assert np.isinf(r["cost"][~ok]).all()
        costs = [float(v) for v in r["cost"]]
        assert costs.index(min(costs)) == int(c["direction_flag"]), name
        seen_ok += int(ok.sum()); seen_bad += int((~ok).sum())
    assert seen_ok >= 8 and seen_bad >= 4


def test_nonconvex_crosscheck(orc, AB, golden_mpccbf, golden_planner):
    """The non-convex goldens (control.mpccbf / control.mpc_multi_agents NLPs) against the independent cross-check recorded
    by tests/golden/tools/crosscheck.py in the build container (nonconvex_crosscheck.npz, numbers only): for each of the
    17 recorded NLPs the golden solver from 8 random dynamically-consistent starts and scipy SLSQP on the condensed
    problem from the reference's own zero start.  Pins: (i) every certified end point of every start IS the golden point
    -- one KKT point per NLP, so "which local solution would IPOPT return from the zero start" has one candidate; (ii) no
    feasible point below the golden cost was ever found; (iii) the oracle's cost is that golden cost."""
    import os

    import conftest

    z = np.load(os.path.join(conftest.GOLDEN, "nonconvex_crosscheck.npz"), allow_pickle=False)
    names = [str(n) for n in z["names"]]
    assert len(names) == 17
    same_slsqp = 0
    A, B = AB
    for n in names:
        assert bool(z[n + "/golden_certified"]), n
        assert int(z[n + "/n_random_certified"]) >= 7, n
        assert z[n + "/distinct_costs"].shape == (1,), (n, z[n + "/distinct_costs"])          # a single KKT point was found
        assert bool(z[n + "/golden_is_global_among_found"]) and bool(z[n + "/golden_is_lowest_from_zero_start"]), n
        assert bool(z[n + "/slsqp_condensed_not_below_golden"]), n
        same_slsqp += int(bool(z[n + "/slsqp_condensed_same_point"]))
        kind, case = n.split("/")
        if kind == "mpccbf":
            g = golden_mpccbf.case(case)
            d, args = helpers.mpccbf_inputs(g, A, B)
        else:
            g = golden_planner.case(case)
            d, args = helpers.mma_inputs(g, A, B)
        r = orc.cbf_solve(_with_tol(d, 1e-11), *args)
        assert r["status"][0] == 0
        assert _close_cost(r["cost"][0], float(z[n + "/golden_cost"]), 1e-9), (n, r["cost"][0], float(z[n + "/golden_cost"]))
    assert same_slsqp >= 12, same_slsqp      # SLSQP stalls early on the rest (higher cost, never lower)


def test_lmpc_prep_regression_and_safe_set(orc, golden_racing_game):
    """crx_oracle_lmpc_prep (local LTV regression + kinematic linearisation + safe-set selection) against the stage models
    and safe-set points the reference itself computed in its learning-MPC lap (tests/golden/racing_game.npz), replayed call
    by call exactly as LMPCRacingGame does: linearisation points from the previous (golden) solution, safe set extended by
    add_point.  The reference's normal matrices reach cond 3e11, so coefficients are pinned loosely (any two LAPACK routes
    differ by ~1e-5 relative) and what the models PREDICT at their own query points tightly; the closed-form kinematic rows
    and the selection (indices, cost-to-go) exactly."""
    import os

    import conftest
    from utils import racing_env

    g = golden_racing_game
    track = racing_env.ClosedTrack(np.genfromtxt(os.path.join(conftest.ROOT, "data/track_layout/l_shape.csv"), delimiter=","), track_width=1.0)
    d, ss, us, qf, time_ss, lin_points, lin_input = helpers.lmpc_lap_setup(g, track)
    N, L = d.N, float(g["lap_length"])
    it = np.array([2], dtype=np.int32)
    worst_c = worst_p = 0.0
    for c in range(int(g["lmpc_first_uncertified"]) + 1):
        r = orc.lmpc_prep(d, ss[None], us[None], qf[None], time_ss[None], it, g["lmpc/x"][c][None], lin_points[None], lin_input[None],
                          track.point_and_tangent)
        assert r["status"][0] == 0
        Ag, Bg, Cg = g["lmpc/A"][c], g["lmpc/B"][c], g["lmpc/C"][c]
        np.testing.assert_allclose(r["A"][0][:, 3:], Ag[:, 3:], atol=1e-12)          # kinematic rows: closed form
        np.testing.assert_allclose(r["C"][0][:, 3:], Cg[:, 3:], atol=1e-12)
        assert (r["B"][0][:, 3:] == 0).all()
        scale = max(1.0, np.abs(Ag).max())
        worst_c = max(worst_c, np.abs(r["A"][0] - Ag).max() / scale, np.abs(r["B"][0] - Bg).max() / scale)
        assert np.abs(r["A"][0] - Ag).max() <= (1e-8 if c < 2 else 2e-5) * scale, c
        pred = np.einsum("nij,nj->ni", r["A"][0], lin_points[:N]) + np.einsum("nij,nj->ni", r["B"][0], lin_input) + r["C"][0]
        pred_g = np.einsum("nij,nj->ni", Ag, lin_points[:N]) + np.einsum("nij,nj->ni", Bg, lin_input) + Cg
        worst_p = max(worst_p, np.abs(pred - pred_g).max())
        np.testing.assert_allclose(pred, pred_g, atol=1e-5)
        np.testing.assert_array_equal(r["ss"][0], g["lmpc/ss"][c])                   # the points the reference put into its QP
        np.testing.assert_array_equal(r["qfun"][0], g["lmpc/qfun"][c])
        # next call: plan shifted by one stage (control.py:726-728), safe set extended (utils/base.py:624-629)
        X, U = g["lmpc/X"][c], g["lmpc/U"][c]
        r2 = orc.lmpc_prep(d, ss[None], us[None], qf[None], time_ss[None], it, g["lmpc/x"][c][None], X[None], U[None],
                           track.point_and_tangent, from_plan=True)
        lin_points, lin_input = np.concatenate((X[1:], X[-1:]), axis=0), np.vstack((U[1:], U[-1]))
        r3 = orc.lmpc_prep(d, ss[None], us[None], qf[None], time_ss[None], it, g["lmpc/x"][c][None], lin_points[None], lin_input[None],
                           track.point_and_tangent)
        for k in ("A", "B", "C"):
            np.testing.assert_array_equal(r2[k], r3[k])                                # from_plan = the shift done inside
        row = time_ss[1] + c + 1
        ss[1, row] = g["lmpc/x"][c] + np.array([0, 0, 0, 0, L, 0])
        us[1, row] = U[0]
    assert worst_c <= 2e-5 and worst_p <= 1e-5, (worst_c, worst_p)


def test_planner_verdicts_against_an_lp_solver(orc, AB):
    """Every verdict of the oracle on 1024 region QPs of the BASELINE cfg3 draw against HiGHS (scipy.optimize.linprog) on
    the QP's feasible set: condensed inputs in their box, vx_k <= vx_max (k >= 1), ey_lb[k] <= ey_k <= ey_ub (1 <= k < N)
    (overtake_traj_planner.py:276-324).  Infeasible verdicts come from box_certificate() (a Farkas proof over the input
    box, found after ~3 iterations) or from the divergence test; "infeasible" must mean infeasible, because the verdict
    selects the reference's fall-back trajectory (:365-374)."""
    from scipy.optimize import linprog
    from crx import abi, synth
    A, B = AB
    N = 12
    p = synth.cfg3_planner(256, N=N, seed=11)
    d = abi.planner_desc(N, A, B)
    r = orc.planner_solve(d, *[p[k] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")])
    st, it = np.asarray(r["status"]), np.asarray(r["iters"])
    # x_k = A^k x0 + sum_j A^(k-1-j) B u_j
    Ap = [np.eye(6)]
    for _ in range(N):
        Ap.append(A @ Ap[-1])
    G = np.zeros((N + 1, 6, 2 * N))
    for k in range(1, N + 1):
        for j in range(k):
            G[k][:, 2 * j:2 * j + 2] = Ap[k - 1 - j] @ B
    bounds = [(-d.delta_max, d.delta_max), (-d.a_max, d.a_max)] * N
    wrong, n_inf, margin = [], 0, 1e-7
    for b in range(len(st)):
        x0, lb, ub = p["x0"][b], p["ey_lb"][b], p["ey_ub"][b]
        free = [Ap[k] @ x0 for k in range(N + 1)]
        rows, rhs = [], []
        for k in range(1, N + 1):
            rows.append(G[k][0]); rhs.append(d.vx_max - free[k][0])
            if k < N:
                if np.isfinite(ub):
                    rows.append(G[k][5]); rhs.append(ub - free[k][5])
                if np.isfinite(lb[k]):
                    rows.append(-G[k][5]); rhs.append(free[k][5] - lb[k])
        lp = linprog(np.zeros(2 * N), A_ub=np.array(rows), b_ub=np.array(rhs), bounds=bounds, method="highs")
        infeas0 = x0[5] < lb[0] - 1e-8 or x0[5] > ub + 1e-8      # a row on the fixed x0 (quirk: IPOPT cannot succeed)
        lp_infeasible = lp.status == 2
        n_inf += int(lp_infeasible or infeas0)
        if (st[b] != 0) != (lp_infeasible or infeas0):
            # a feasible set thinner than the solvers' tolerances may fall either way: re-test with the rows moved by `margin`
            lo = linprog(np.zeros(2 * N), A_ub=np.array(rows), b_ub=np.array(rhs) - margin, bounds=bounds, method="highs").status == 2
            hi = linprog(np.zeros(2 * N), A_ub=np.array(rows), b_ub=np.array(rhs) + margin, bounds=bounds, method="highs").status == 2
            if lo == hi:
                wrong.append((b, int(st[b]), int(it[b]), int(lp.status)))
    assert not wrong, wrong
    assert 300 < n_inf < 600                                   # ~41 % of the draw, as BASELINE's cfg3 recipe produces them
    assert it[st != 0].mean() < 5.0 and it[st != 0].max() <= 20   # the proof is found early
    # [r4] CRX_INFEASIBLE is a PROOF (screen, certificate, a row on the fixed x0); the divergence heuristic reports CRX_STALLED:
    # on this draw every failed QP is a proved one
    assert (st[st != 0] == 2).all(), np.bincount(st)
    # [r3] the reachability screen (crx_ipm_opts.reach_screen) answers before the first iteration: it must flag nothing the
    # interior-point route (multiplier certificate) solves, and it catches the regions the bicycle cannot reach --
    # on this draw nearly all of the infeasible ones
    screened = (st == 2) & (it == 0)
    d0 = abi.planner_desc(N, A, B, opts=abi.default_opts(reach_screen=0))
    r0 = orc.planner_solve(d0, *[p[k] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")])
    st0, it0 = np.asarray(r0["status"]), np.asarray(r0["iters"])
    np.testing.assert_array_equal(st0 != 0, st != 0)             # the same verdict either way
    assert (st0[st0 != 0] == 2).mean() >= 0.99                   # ... proved by the multiplier certificate instead (a rare one ends by the divergence heuristic: CRX_STALLED)
    assert (it0[screened] >= 1).all() and (it0[~screened] == it[~screened]).all()
    np.testing.assert_array_equal(np.asarray(r0["X"]), np.asarray(r["X"]))   # and the same trajectory (fall-back for the failed ones)
    assert screened.sum() >= 0.9 * (st != 0).sum(), (int(screened.sum()), int((st != 0).sum()))


def test_cbf_slack_start_oracle(orc, AB):
    """Oracle side of crx_ipm_opts.slack_start: 0 = the reference's zero start + closed-form restoration (libcrx 0.1.x), 1 = slacks at
    their provable lower bounds, 2 (default) = the crash path.  Problems that never enter a non-default path keep their bits; each
    step up converges more crash states, and the crash path converges ALL of this draw in fewer iterations."""
    from crx import abi, synth
    A, B = AB
    p = synth.cfg2_mpccbf(512, N=12, seed=2, safe_start=False)
    args = [p[k] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")]
    r = {ss: orc.cbf_solve(abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"], opts=abi.default_opts(slack_start=ss)), *args) for ss in (0, 1, 2)}
    r0 = r[0]
    n0 = (r0["status"] == 0).sum()
    assert set(np.unique(r0["status"])) <= {0, 3, 5}            # nothing is called infeasible without a proof
    for ss in (1, 2):
        r1 = r[ss]
        touched = (np.abs(r1["X"] - r0["X"]).reshape(512, -1).max(axis=1) > 0) | (r1["iters"] != r0["iters"]) | (r1["status"] != r0["status"])
        assert 8 <= touched.sum() <= 80, touched.sum()
        assert (r1["status"] == 0).sum() >= n0 + 8
        both = (r0["status"] == 0) & (r1["status"] == 0)
        rel = np.abs(r1["cost"][both] - r0["cost"][both]) / np.maximum(1.0, np.abs(r0["cost"][both]))
        assert (rel <= 1e-6).mean() >= 0.98                                          # the same KKT point where both converge
    conv2 = r[2]["status"] == 0
    assert conv2.mean() >= 0.995 and conv2.sum() > (r[1]["status"] == 0).sum(), (conv2.sum(), n0)   # the crash path: (nearly) every NLP of the draw
    assert np.percentile(r[2]["iters"], 99) <= 35 < np.percentile(r0["iters"], 99)   # ... and the tail is shorter
    assert r[2]["kkt"][conv2].max() <= 1e-8
    # [r4b] 3 = the eager crash path: every violated zero start takes the candidate point (no restart) -- the whole draw, a shorter tail still;
    # problems whose zero start violates nothing keep their bits
    r3 = orc.cbf_solve(abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"], opts=abi.default_opts(slack_start=3)), *args)
    assert (r3["status"] == 0).mean() >= 0.995 and r3["iters"].max() <= r[2]["iters"].max() and np.percentile(r3["iters"], 99) <= np.percentile(r[2]["iters"], 99)
    same = (np.abs(r3["X"] - r[2]["X"]).reshape(512, -1).max(axis=1) == 0) & (r3["iters"] == r[2]["iters"])
    assert 0.85 <= same.mean() < 1.0, same.mean()


def test_lmpc_reach_screen_oracle(orc, golden_racing_game):
    """Oracle side of the learning-MPC reachability screen: skipping a first attempt that is provably infeasible changes
    nothing but the iteration count."""
    import helpers
    d, args = helpers.lmpc_inputs(golden_racing_game)
    on = orc.lmpc_solve(d, *args)
    import copy
    d_off = copy.deepcopy(d)
    d_off.opts.reach_screen = 0
    off = orc.lmpc_solve(d_off, *args)
    for k in ("status", "X", "U", "lam", "cost"):
        np.testing.assert_array_equal(on[k], off[k])
    fewer = on["iters"] < off["iters"]
    assert (on["iters"] <= off["iters"]).all() and fewer.sum() >= 4 and (on["status"][fewer] == 2).all()
    assert (on["status"][golden_racing_game["lmpc_success"]] == 0).all()     # no solvable QP is touched


def test_lmpc_noise_floor_qps_end_early(orc):
    """tests/golden/lmpc_noise_floor.npz (see tests/test_gpu_parity.py::test_lmpc_noise_floor_qps): the oracle leaves these
    QPs after < 80 iterations with a status != 0, not after max_iter."""
    import os

    import conftest
    from crx import abi
    z = np.load(os.path.join(conftest.ROOT, "tests", "golden", "lmpc_noise_floor.npz"))
    r = orc.lmpc_solve(abi.lmpc_desc(12, 44), *[z[k] for k in ("x0", "u_old", "A", "B", "C", "ss", "qfun", "n_ss")])
    assert (r["status"] != 0).all() and r["iters"].max() <= 80 and np.isfinite(r["U"]).all()


def _oracle_addtraj(lib, d, crossed, log_x, log_u, n_log, ss, us, qf, time_ss, it, step, x):
    import ctypes as C
    status = np.zeros(len(crossed), dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    rc = lib.crx_oracle_lmpc_addtraj(C.byref(d), C.c_int(len(crossed)), p(crossed), p(log_x), p(log_u), p(n_log), p(ss), p(us), p(qf),
                                     p(time_ss), p(it), p(step), p(x), p(status))
    assert rc == 0
    return status


def test_lmpc_add_trajectory_vs_reference_safe_set(golden_racing_game):
    """crx_oracle_lmpc_addtraj (LMPCRacingGame.add_trajectory, utils/base.py:631-656) against the safe set the reference's
    own code built from its PID lap and its mpc-lti lap (tests/golden/racing_game.npz: lap0/*, lap1/* are the logged laps,
    ss/* the arrays after both add_trajectory calls): states, inputs, time_ss and the whole cost-to-go column incl. the
    reference's count-down pass -- exact."""
    import ctypes as C

    import oracle
    from crx import abi
    oracle.load()
    lib = C.CDLL(oracle._LIB)
    g = golden_racing_game
    P, L = g["ss/ss0"].shape[0], g["ss/ss0"].shape[2]
    d = abi.lmpcprep_desc(12, P, L, 9, float(g["timestep"]), float(g["lap_length"]))
    ss, us, qf = np.zeros((1, L, P, 6)), np.zeros((1, L, P, 2)), np.zeros((1, L, P))
    time_ss, it, step = np.zeros((1, L), dtype=np.int32), np.zeros(1, dtype=np.int32), np.full(1, 7, dtype=np.int32)
    for lap in (0, 1):
        xs, u = g["lap%d/xcurv" % lap], g["lap%d/u" % lap]
        log_x, log_u = np.zeros((1, P, 6)), np.zeros((1, P, 2))
        log_x[0, :len(xs)] = xs; log_u[0, :len(u)] = u
        n_log = np.array([len(xs)], dtype=np.int32)
        x_next = xs[-1] - np.array([0, 0, 0, 0, float(g["lap_length"]), 0])
        st = _oracle_addtraj(lib, d, np.ones(1, dtype=np.int32), log_x, log_u, n_log, ss, us, qf, time_ss, it, step, x_next[None].copy())
        assert st[0] == 0 and it[0] == lap + 1 and step[0] == 0 and n_log[0] == 1
        np.testing.assert_array_equal(log_x[0, 0], x_next)
        step[0] = 5
    ref_ss = np.ascontiguousarray(g["ss/ss0"].transpose(2, 0, 1)); ref_u = np.ascontiguousarray(g["ss/u0"].transpose(2, 0, 1))
    ref_q = np.ascontiguousarray(g["ss/Qfun0"].T)
    for lap in (0, 1):
        n = int(g["ss/time_ss"][lap])
        assert time_ss[0, lap] == n == len(g["lap%d/u" % lap])
        np.testing.assert_array_equal(ss[0, lap, :n + 1], ref_ss[lap, :n + 1])
        np.testing.assert_array_equal(us[0, lap, :n], ref_u[lap, :n])
        np.testing.assert_array_equal(qf[0, lap], ref_q[lap])                   # the whole column, count-down included
    # a race that did not cross is left alone; a full safe set reports status 1 and restarts the log
    it[0] = L
    log_x = np.ones((1, P, 6)); n_log = np.array([40], dtype=np.int32); keep = ss.copy()
    st = _oracle_addtraj(lib, d, np.zeros(1, dtype=np.int32), log_x, np.ones((1, P, 2)), n_log, ss, us, qf, time_ss, it, step, np.zeros((1, 6)))
    assert st[0] == 0 and n_log[0] == 40 and (ss == keep).all()
    st = _oracle_addtraj(lib, d, np.ones(1, dtype=np.int32), log_x, np.ones((1, P, 2)), n_log, ss, us, qf, time_ss, it, step, np.zeros((1, 6)))
    assert st[0] == 1 and n_log[0] == 1 and it[0] == L and (ss == keep).all()


@pytest.mark.parametrize("eps", [1e-3, 1e-6, 1e-8])
def test_certificate_never_fires_on_razor_thin_feasible_qps(orc, AB, eps):
    """The infeasibility proof on planner QPs that are feasible by a hair: the ey corridor is a tube of half-width eps
    around a known feasible trajectory.  Every one of them must converge (none may be declared infeasible)."""
    d, args = helpers.thin_corridor_qps(orc, AB, eps)
    r = orc.planner_solve(d, *args)
    assert len(r["status"]) > 200 and (np.asarray(r["status"]) == 0).all(), np.bincount(np.asarray(r["status"]), minlength=3)


def test_certificate_fires_on_qps_infeasible_by_a_hair(orc, AB):
    """... and with the tube turned inside out by 1e-6 every one of them is infeasible, and is reported so."""
    d, args = helpers.thin_corridor_qps(orc, AB, -1e-6)
    r = orc.planner_solve(d, *args)
    st = np.asarray(r["status"])
    assert np.isin(st, (2, 5)).all() and (st == 2).mean() >= 0.99, np.bincount(st, minlength=6)   # proved (2); a rare one by the divergence heuristic (5)


def test_plant_step_with_process_noise(orc):
    """crx_oracle_plant_step_noise against the reference's own DynamicBicycleModel.forward_dynamics WITH its bounded process
    noise (utils/base.py:929-939; tests/golden/plant_noise.npz, make_draws.py plant_noise): same standard-normal draws in,
    same state out -- half of the clipped noise on the curvilinear velocities, none on the global-frame copy."""
    import ctypes as C
    import os

    import conftest
    from crx import abi

    g = np.load(os.path.join(conftest.GOLDEN, "plant_noise.npz"))
    n = len(g["seed"])
    d = abi.plant_desc(g["table"].shape[0], float(g["lap_length"]))
    xg, xc = np.zeros((n, 6)), np.zeros((n, 6))
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (g["table"], g["xglob"], g["xcurv"], g["u"], g["z"])]
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    fn = orc.lib.crx_oracle_plant_step_noise
    fn.restype = C.c_int
    assert fn(C.byref(d), C.c_int(n), *[p(a) for a in arrs], p(xg), p(xc)) == 0
    np.testing.assert_allclose(xc, g["xcurv_next"], rtol=0, atol=2e-13)
    np.testing.assert_allclose(xg, g["xglob_next"], rtol=0, atol=2e-13)
    clipped = np.abs(g["z"] * np.array([0.01, 0.01, 0.005])) > np.array([0.05, 0.1, 0.05])
    assert clipped.sum() >= 3 and (~clipped).sum() >= 30                      # both sides of every clip are exercised
    assert np.abs(xc[:, :3] - xg[:, :3]).max() > 1e-3                          # the noise is there, on xcurv only
    # zero noise = the plain step
    r0 = orc.plant_step(d, g["table"], g["xglob"], g["xcurv"], g["u"])
    np.testing.assert_array_equal(r0["xglob"], xg)
