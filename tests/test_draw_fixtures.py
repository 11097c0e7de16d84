"""The BASELINE draws bench.py times, tied to the reference's OWN problem construction.

tests/golden/cfg{2,3,4}_draw.npz (tests/golden/tools/make_draws.py) replay the synthetic scenarios of crx/synth.py through
the reference's unmodified control.mpccbf / mpc_multi_agents / OvertakeTrajPlanner.get_local_traj under the recording CasADi
stand-in.  Three-way checks, CPU side here (the `-m gpu` twins are in tests/test_gpu_parity.py):

  rows      cost, variable boxes and CBF row values of the reference's recorded problem at a seeded point == the rows
            oracle/crx_oracle.c builds from the arrays the PRODUCT's host prep (crx/synth.py + crx/hostprep.py, the path
            bench.py takes) makes of the same raw scenario -- for every problem, crash states included, no solve involved;
  prep      obstacles that enter the NLP, their predictions, Bezier polylines, sorted vehicles == the reference's;
  solutions oracle == certified KKT point of the reference-built problem (third solver, or the oracle's own point put
            through the solver-agnostic certificate on the reference's recorded graph: `how`), direction flags exact;
  classes   every problem the oracle does not converge on is classified: `third solver finds no KKT point either` /
            `a KKT point exists (solvable; the oracle / kernel stop for latency or by their restoration budget)`.
"""
import os

import numpy as np
import pytest

import conftest
import helpers

LAP = 19.22957795362994


def _load(name):
    path = os.path.join(conftest.GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(name + " not generated")
    return np.load(path, allow_pickle=False)


def _group(z, g):
    return {k[len(g) + 1:]: z[k] for k in z.files if k.startswith(g + "/")}


def _rel(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def cbf_batch(kind, lapped):
    from crx import synth
    if kind == "cfg2":
        return synth.cfg2_mpccbf(256, N=12, seed=2, safe_start=False, lapped_frac=0.25 if lapped else 0.0)
    return synth.cfg4_tracking_cbf(256, N=20, seed=4, safe_start=False, lapped_frac=0.25 if lapped else 0.0)


def cbf_desc(kind, A, B, tol=1e-8):
    from crx import abi
    if kind == "cfg2":
        d = abi.cbf_desc(12, 1, A, B, alpha=0.8, margin=0.2)
    else:
        d = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    d.opts.tol = tol
    return d


KEYS = ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")


@pytest.mark.parametrize("kind,group", [("cfg2", "draw"), ("cfg2", "lapped"), ("cfg4", "draw"), ("cfg4", "lapped")])
def test_cbf_rows_and_prep_match_reference(orc, AB, kind, group):
    """synth + hostprep build the reference's rows (module docstring: rows, prep)."""
    from crx import hostprep
    A, B = AB
    g = _group(_load(kind + "_draw.npz"), group)
    p = cbf_batch(kind, group == "lapped")
    d = cbf_desc(kind, A, B)
    N, V = d.N, d.n_obs_max
    assert len(g["index"]) >= (12 if group == "lapped" else 48) or os.environ.get("CRX_DRAW_PARTIAL")
    n_lapped = 0
    for r, b in enumerate(g["index"]):
        b = int(b)
        tag = "%s/%s #%d" % (kind, group, b)
        # the fixture replayed exactly what synth draws today
        np.testing.assert_array_equal(p["x0"][b], g["x0"][r], err_msg=tag)
        np.testing.assert_array_equal(p["cars"][b], g["cars"][r], err_msg=tag)
        n = int(g["n_obs_ref"][r])
        assert int(p["n_obs"][b]) == n, tag                                   # the window test kept the same cars ...
        keep, off = hostprep.cbf_window(p["x0"][b:b + 1], g["cars"][r][None, :, 0], LAP)
        kept = np.nonzero(keep[0])[0]
        assert len(kept) == n, tag
        # ... and their predictions are what the reference's vehicle model returned (sympy s(t) = v t + s0 vs numpy)
        np.testing.assert_allclose(p["obs_s"][b, :n], g["obs_pred"][r][kept, 4, :], rtol=0, atol=2e-14, err_msg=tag)
        np.testing.assert_allclose(p["obs_ey"][b, :n], g["obs_pred"][r][kept, 5, :], rtol=0, atol=1e-15, err_msg=tag)
        n_lapped += bool(np.any(p["lap_off"][b, :n] != 0.0))
        # rows at the probe point
        U, sig = g["probe_U"][r], g["probe_sigma"][r][:n]
        pr = helpers.oracle_cbf_probe(orc, d, p["x0"][b], p["xt"][b], p["obs_s"][b], p["obs_ey"][b], p["lap_off"][b], n, U, sig)
        assert g["probe_eq"][r] <= 1e-12, tag                                # dynamics + x0 rows vanish on the roll-out: same A, B, x0
        assert _rel(pr["cost"], g["probe_f"][r]) <= 1e-12, (tag, pr["cost"], g["probe_f"][r])
        ref_rows = g["probe_cbf"][r][:n]
        assert (_rel(pr["cbf"], ref_rows) <= 1e-9).all(), (tag, np.abs(pr["cbf"] - ref_rows).max())
        # boxes: inputs, vx and ey of every stage (the reference also states them on the fixed x0), sigma >= 0
        # (the fixture recovers a bound as -(c(z) - a z) / a from the recorded row: exact up to the rounding of that expression)
        bx = dict(rtol=0, atol=4e-15, err_msg=tag)
        for c in range(2):
            np.testing.assert_allclose(g["box_u_lo"][r][:, c], pr["ulo"][c], **bx)
            np.testing.assert_allclose(g["box_u_hi"][r][:, c], pr["uhi"][c], **bx)
        np.testing.assert_allclose(g["box_x_lo"][r][:, 0], pr["vlo"], **bx)
        np.testing.assert_allclose(g["box_x_hi"][r][:, 0], pr["vhi"], **bx)
        np.testing.assert_allclose(g["box_x_lo"][r][:, 5], pr["elo"], **bx)
        np.testing.assert_allclose(g["box_x_hi"][r][:, 5], pr["ehi"], **bx)
        assert np.isinf(g["box_x_lo"][r][:, 1:5]).all() and np.isinf(g["box_x_hi"][r][:, 1:5]).all(), tag
        assert (np.abs(g["box_sig_lo"][r][:n]) <= 4e-15).all(), tag
    if group == "lapped":
        assert n_lapped >= len(g["index"]) // 2, "the lapped draw must exercise lap_off != 0 (quirk Q1)"


# cfg4 (N = 20, three obstacles, reduced Hessian cond 1e8): measured worst over the 48 + 12 problems -- tol 1e-8: x 2.9e-4, vx/ey 3e-6,
# u 1.8e-4, cost 3.2e-7; tol 1e-11: x 5.2e-6, vx/ey 2.5e-6, u 4.2e-5, cost 1.8e-9
TOL4 = {1e-8: dict(x=5e-4, u=2e-3, f=1e-6, xw=1e-4), 1e-11: dict(x=2e-5, u=1e-4, f=5e-9, xw=5e-6)}


def _solve_and_compare(binding, orc, AB, kind, group, tol, T):
    """Shared by the CPU (oracle) and GPU (libcrx) versions.  Returns the classification table."""
    A, B = AB
    if kind == "cfg4":
        T = dict(tol=tol, n_loose=0, **TOL4[tol])
    g = _group(_load(kind + "_draw.npz"), group)
    p = cbf_batch(kind, group == "lapped")
    d = cbf_desc(kind, A, B, tol)
    idx = g["index"].astype(int)
    res = binding.cbf_solve(d, *[p[k][idx] for k in KEYS])
    N = d.N
    how = g["how"] if "how" in g else np.where(g["success"], 0, np.where(g["retry_certified"], 1, -1))
    # [r4] a second certified KKT point of the same reference-built NLP where the crash path (slack_start = 2) ends elsewhere than the
    # primary point (make_draws.py alt_pass): the solve must land on ONE of the certified points
    alt_ok = g["alt_ok"] if "alt_ok" in g else np.zeros(len(idx), dtype=bool)
    table = dict(converged_certified=0, converged_uncertified=0, stopped_solvable=0, stopped_unsolved=0, other_kkt_point=0, alt_point=0)
    worst = dict(x=0.0, xw=0.0, u=0.0, f=0.0)
    loose = []
    for r, b in enumerate(idx):
        tag = "%s/%s #%d how %d status %d iters %d" % (kind, group, b, how[r], res["status"][r], res["iters"][r])
        conv, cert = res["status"][r] == 0, how[r] >= 0 or bool(alt_ok[r])
        table["converged_certified" if conv and cert else "converged_uncertified" if conv else "stopped_solvable" if cert else "stopped_unsolved"] += 1
        if not (conv and cert):
            continue
        n = int(g["n_obs_ref"][r])
        gX, gU, gS, fg = g["X"][r], g["U"][r], g["sigma"][r], g["cert"][r][0]
        if alt_ok[r]:
            fa = g["alt_cert"][r][0]
            if how[r] < 0 or abs(res["cost"][r] - fa) / max(1.0, abs(fa)) < abs(res["cost"][r] - fg) / max(1.0, abs(fg)):
                gX, gU, gS, fg = g["alt_X"][r], g["alt_U"][r], g["alt_sigma"][r], fa
                table["alt_point"] += 1
        df = abs(res["cost"][r] - fg) / max(1.0, abs(fg))
        # non-convex rows: the point must be the certified one unless it is another KKT point that is no worse
        if df > T["f"] and res["cost"][r] < fg:
            table["other_kkt_point"] += 1
            continue                                                          # a better local minimum than the third solver's: allowed
        assert df <= T["f"], (tag, res["cost"][r], fg)
        dX, dU = np.abs(res["X"][r] - gX), np.abs(res["U"][r] - gU)
        scale = max(1.0, abs(fg) * 1e-6)                                      # crash states: costs 1e8..1e12, slacks 1e4..1e8
        if abs(fg) > 1e5:    # [r4] crash states reach their point along the crash path, not along the path the stored point was reached by:
            scale *= 4.0     # the uncosted states (vy, wz) of such a point are only determined to ~1e-2 at tol 1e-8 (#86: wz 9.7e-3, cost to 1e-9)
        # cost-weighted states of both NLPs: vx, ey (Q = diag(10,0,0,4|5,0,40|50): s carries no cost and is only as sharp as x)
        Tl = T.get("loose", T)
        assert dX.max() <= Tl["x"] * scale and dX[:, [0, 5]].max() <= Tl["xw"] * scale, (tag, dX.max(), dX[:, [0, 5]].max())
        assert dU.max() <= Tl["u"] * scale, (tag, dU.max())
        if not (dX.max() <= T["x"] * scale and dX[:, [0, 5]].max() <= T["xw"] * scale and dU.max() <= T["u"] * scale):
            loose.append(tag)
        if n:
            ds = np.abs(res["sigma"][r][:n] - gS[:n])
            assert (ds <= 1e-6 * np.maximum(1.0, np.abs(gS[:n]))).all(), (tag, ds.max())
        for k, v in (("x", dX.max()), ("xw", dX[:, [0, 5]].max()), ("u", dU.max()), ("f", df)):
            worst[k] = max(worst[k], v / (scale if k != "f" else 1.0))
    return table, worst, res, g, how


# tol 1e-11 on both sides: 252 of the 256 + 32 cfg2 problems that have a certified point agree to x 4e-7 / u 6e-6 (vx, ey 5e-7);
# ONE (#131) sits on a weakly determined direction and agrees to 1.8e-5 only (cost to 1e-11) -- `loose` bounds every problem,
# the strict set all but `n_loose` of them.  Four more (#6, #33, #95, #163) end at a DIFFERENT KKT point of lower cost than the
# third solver's from the same zero start (non-convex NLP): listed by the test, not compared.
TIGHT = dict(tol=1e-11, x=1e-6, u=1e-5, f=1e-9, xw=1e-6, loose=dict(x=5e-5, u=5e-5, xw=2e-6), n_loose=2)
# tol 1e-8 against points solved to 1e-11: on problems with an ACTIVE CBF row (most of these draws; few of the hand-made
# goldens) the barrier perturbation mu ~ 1e-9 moves vx / ey by up to ~3e-5 (the row's gradient is 1e2..1e9)
DEFAULT = dict(tol=1e-8, x=5e-4, u=2e-3, f=1e-7, xw=1e-4)


@pytest.mark.parametrize("T", [DEFAULT, TIGHT], ids=["tol1e-8", "tol1e-11"])
@pytest.mark.parametrize("kind,group", [("cfg2", "draw"), ("cfg2", "lapped"), ("cfg4", "draw"), ("cfg4", "lapped")])
def test_oracle_solutions_on_reference_built_draws(orc, AB, kind, group, T):
    table, worst, res, g, how = _solve_and_compare(orc, orc, AB, kind, group, T["tol"], T)
    print("\n%s/%s tol %g: %s worst %s" % (kind, group, T["tol"], table, {k: "%.1e" % v for k, v in worst.items()}))
    n = len(how)
    assert table["converged_certified"] >= 0.8 * n, table
    # every converged solve must be backed by a certificate on the reference's graph (how = 3 certifies the oracle's own
    # tol-1e-11 point there); a converged, uncertified one would mean the oracle solves a different problem
    # [r6] no exception any more.  cfg2 #99 (cost 1.5e8, multipliers of 1e9) used to be one: "converged" meant the SCALED error alone, its complementarity sat
    # at 1e-5..1e-4, and the active-set multiplier recovery of the certificate (rows active to 1e-7) dropped three rows with slacks of 3e-7..4e-5 and
    # multipliers of 0.2: 0.26..0.86 of "stationarity defect".  With IPOPT's complete termination test in the solver and the certificate's second multiplier
    # recovery (nlp_solve.kkt_certificate_ipopt: a linear program over all rows at IPOPT's complementarity tolerance) the point is certified: 4.5e-7.
    known = 0
    assert table["converged_uncertified"] <= (known if T["tol"] <= 1e-10 else max(1, n // 50)), table


def test_headline_non_converged_are_classified(orc, AB):
    """BENCH's headline batch (cfg2, 256 problems, SURVEY 8d draw, unfiltered): every problem that does not end CONVERGED at the
    product's default options is classified by what is known about the reference-built problem."""
    T = DEFAULT
    table, worst, res, g, how = _solve_and_compare(orc, orc, AB, "cfg2", "draw", T["tol"], T)
    st = res["status"]
    stopped = np.nonzero(st != 0)[0]
    lines = []
    for r in stopped:
        lines.append("#%d status %d iters %d: %s" % (int(g["index"][r]), st[r], res["iters"][r],
                     {-1: "no KKT point known (third solver, 4 random starts and the oracle at 1e-11 all fail)", 0: "KKT point exists (third solver, zero start)",
                      1: "KKT point exists (third solver, 1000 iterations)", 2: "KKT point exists (third solver, random start)",
                      3: "KKT point exists (oracle at tol 1e-11, certified on the reference's graph)"}[int(how[r])]))
    print("\nheadline batch: %d of %d not converged\n  " % (len(stopped), len(st)) + "\n  ".join(lines))
    assert len(st) == 256
    assert len(stopped) <= 3, len(stopped)        # [r4] the crash path: 256 of 256 (VERDICT r3 asked for >= 253); 245 with slack_start = 0


def cfg4_stopped_batch():
    """The problems of tests/golden/cfg4_stopped.npz, taken from the benched configs[3] batch itself (16 384 tracking NLPs, seed 4, unfiltered)."""
    from crx import synth
    g = _group(_load("cfg4_stopped.npz"), "draw")
    p = synth.cfg4_tracking_cbf(16384, N=20, seed=4, safe_start=False)
    idx = g["index"].astype(int)
    for r in (0, len(idx) // 2, len(idx) - 1):
        np.testing.assert_array_equal(p["x0"][idx[r]], g["x0"][r])
    return g, {k: p[k][idx] for k in KEYS}


def classify_cfg4_stopped(res, g):
    """status of every problem next to what is known about the reference's NLP: (lines, counts)"""
    how = np.where(g["success"], 0, np.where(g["retry_certified"], 1, -1))
    known_f = np.where(g["success"], g["cert"][:, 0], g["retry_f"])
    names = {1: "no acceptable step at a point inside the constraints (reported CRX_MAX_ITER)", 3: "restored (budget used up, feasible through its slacks)",
             5: "stalled"}
    known = {0: "a certified KKT point exists (third solver from the zero start, cost %.6g)", 1: "a certified KKT point exists (third solver, 1000 iterations, cost %.6g)"}
    lines, counts = [], dict(stopped=0, stopped_solvable=0, stopped_unknown=0, converged=0, same_point=0, lower_cost=0, higher_cost=0, nothing_to_compare=0)
    for r in range(len(how)):
        st, f = int(res["status"][r]), float(res["cost"][r])
        what = known[int(how[r])] % known_f[r] if how[r] >= 0 else "NO KKT point known (third solver: zero start, then 1000 iterations)"
        if st == 0:
            counts["converged"] += 1
            rel = (f - known_f[r]) / max(1.0, abs(known_f[r])) if how[r] >= 0 else np.nan
            cls = "nothing_to_compare" if how[r] < 0 else "same_point" if abs(rel) <= 1e-6 else "lower_cost" if rel < 0 else "higher_cost"
            counts[cls] += 1
            lines.append("#%d CONVERGED after %d iterations, cost %.6g (%s): %s" % (int(g["index"][r]), int(res["iters"][r]), f, cls.replace("_", " "), what))
            continue
        counts["stopped"] += 1
        counts["stopped_solvable" if how[r] >= 0 else "stopped_unknown"] += 1
        lines.append("#%d status %d (%s) after %d iterations: %s" % (int(g["index"][r]), st, names.get(st, "?"), int(res["iters"][r]), what))
    return lines, counts


def test_cfg4_non_converged_are_classified(orc, AB):
    """BASELINE configs[3] (tracking NLP, N = 20, three cars) left 0.8 % of its 16 384 problems not converged at round 4's default
    options (108 restored, 18 stalled, 7 without an acceptable step on the GPU; 111 / 18 / 6 on the oracle).  Every stalled / stepless one and a third
    of the restored ones were replayed through the reference's own control.mpc_multi_agents (tests/golden/tools/make_draws.py cfg4stopped) and handed
    to the third solver: does the reference's NLP have a KKT point that libcrx missed?  It has, for 46 of the 61 -- and with the restoration phase off
    the oracle itself converges on 48 of them after 32..86 iterations: the 50-iteration stall rule (calibrated on configs[1], whose healthy solves are
    done after 30) and the 25-iteration restoration budget had stopped healthy solves.  Round 5's defaults (stall rule at 100 iterations,
    restore_iters = 50) are the answer; this test pins (1) the old behaviour (oracle knob 2 = 50, restore_iters = 25: none of the 61 converges), (2) the
    new one (at least 52 converge; where the third solver's point has the same cost it is the same point), (3) that the rows the product's prep
    builds equal the reference's on every one of them (probe of cost and CBF rows at a seeded point), crash states included."""
    import ctypes
    A, B = AB
    g, p = cfg4_stopped_batch()
    d = cbf_desc("cfg4", A, B, DEFAULT["tol"])
    assert d.opts.restore_iters == 50
    # (1) round 4's budgets
    d.opts.restore_iters = 25
    orc.lib.crx_oracle_set_knob(2, ctypes.c_double(50.0))
    try:
        old = orc.cbf_solve(d, *[p[k] for k in KEYS])
    finally:
        orc.lib.crx_oracle_set_knob(2, ctypes.c_double(100.0))
    assert (old["status"] != 0).all()
    print("\nconfigs[3], round 4's budgets: restored %d, stalled %d, no acceptable step %d" % tuple(int((old["status"] == s).sum()) for s in (3, 5, 1)))
    # (2) the defaults
    d.opts.restore_iters = 50
    res = orc.cbf_solve(d, *[p[k] for k in KEYS])
    lines, counts = classify_cfg4_stopped(res, g)
    print("configs[3], the same %d problems at the defaults: %s\n  " % (len(g["index"]), counts) + "\n  ".join(lines))
    assert counts["converged"] >= 52 and counts["stopped"] <= 9, counts
    assert res["kkt"][res["status"] == 0].max() <= DEFAULT["tol"]
    how = np.where(g["success"], 0, np.where(g["retry_certified"], 1, -1))
    for r in np.nonzero((res["status"] == 0) & (how == 0))[0]:
        if abs(res["cost"][r] - g["cert"][r][0]) <= 1e-6 * max(1.0, abs(g["cert"][r][0])):
            assert np.abs(res["U"][r] - g["U"][r]).max() <= DEFAULT["u"] and np.abs(res["X"][r] - g["X"][r]).max() <= DEFAULT["x"], int(g["index"][r])
    # (3) the reference's rows at a seeded point = the oracle's rows for the product's arrays (what test_cbf_rows_and_prep_match_reference asserts on the draws)
    for r in range(len(g["index"])):
        n = int(g["n_obs_ref"][r])
        assert n == int(p["n_obs"][r]), r
        pr = helpers.oracle_cbf_probe(orc, d, p["x0"][r], p["xt"][r], p["obs_s"][r], p["obs_ey"][r], p["lap_off"][r], n, g["probe_U"][r], g["probe_sigma"][r][:n])
        assert _rel(pr["cost"], g["probe_f"][r]) <= 1e-10, (r, pr["cost"], g["probe_f"][r])
        assert (_rel(pr["cbf"], g["probe_cbf"][r][:n]) <= 1e-9).all(), r


# ---------------------------------------------------------------------------------------------------------------------
# cfg3: the overtake planner
# ---------------------------------------------------------------------------------------------------------------------
# the BASELINE draw (three cars, seed 3) and [r5] five vehicles of interest per scenario = six regions (CRX_MAX_VEH = 6; the reference plans around
# every vehicle get_overtake_flag returns, overtake_traj_planner.py:62-92): (fixture, V, seed, scenarios at least)
PLANNER_DRAWS = [("cfg3_draw.npz", 3, 3, 48), ("cfg3_many.npz", 5, 35, 20)]
PLANNER_IDS = ["three_cars", "five_cars"]


def planner_batch(V=3, seed=3):
    from crx import synth
    return synth.cfg3_planner(1024, N=12, seed=seed, V=V)


@pytest.mark.parametrize("draw", PLANNER_DRAWS, ids=PLANNER_IDS)
def test_planner_prep_and_rows_match_reference(orc, AB, draw):
    from crx import abi
    A, B = AB
    fixture, V, seed, n_min = draw
    g = _group(_load(fixture), "draw")
    p = planner_batch(V, seed)
    N, R = 12, V + 1
    d = abi.planner_desc(N, A, B)
    n = len(g["index"])
    assert n >= n_min or os.environ.get("CRX_DRAW_PARTIAL")
    assert g["overtake_flag"].all() and (g["n_interest"] == V).all() and (g["n_veh_ref"] == V).all()   # synth's cars are all of interest
    for r, b in enumerate(g["index"].astype(int)):
        tag = "cfg3 #%d" % b
        np.testing.assert_array_equal(p["raw"]["x"][b], g["x"][r], err_msg=tag)
        np.testing.assert_array_equal(p["raw"]["cars"][b], g["cars"][r], err_msg=tag)
        assert int(p["old_flag"][b]) == int(g["old_flag"][r])
        # sorted vehicles (quirk Q3) and their predictions
        from crx import synth
        order = synth.partial_sort_order(g["cars"][r][None, :, 2])[0]
        np.testing.assert_array_equal(order, g["sorted_idx"][r], err_msg=tag)
        np.testing.assert_allclose(p["obs_s"][b], g["obs_pred"][r][:, 4, :], rtol=0, atol=2e-14, err_msg=tag)
        np.testing.assert_allclose(p["obs_ey"][b], g["obs_pred"][r][:, 5, :], rtol=0, atol=1e-15, err_msg=tag)
        sl = slice(b * R, (b + 1) * R)
        # Bezier polylines of every region (planner_helper through the mirror)
        np.testing.assert_allclose(p["bez_s"][sl], g["bezier"][r][:, :, 0], rtol=0, atol=1e-12, err_msg=tag)
        np.testing.assert_allclose(p["bez_ey"][sl], g["bezier"][r][:, :, 1], rtol=0, atol=1e-12, err_msg=tag)
        for reg in range(R):
            q = b * R + reg
            # boxes of the reference's rows: ey_k in [max over its (<= 3) lower-bound rows, w - 0.1] for k < N, vx_k <= 5 for k >= 1
            lo, hi = g["region_box_x_lo"][r][reg], g["region_box_x_hi"][r][reg]
            np.testing.assert_allclose(lo[:N, 5], p["ey_lb"][q], rtol=0, atol=1e-14, err_msg=tag)
            assert (np.abs(hi[:N, 5] - p["ey_ub"][q]) <= 4e-15).all() and np.isinf(lo[N, 5]) and np.isinf(hi[N, 5]), tag
            assert (np.abs(hi[1:, 0] - d.vx_max) <= 4e-15).all() and np.isinf(hi[0, 0]) and np.isinf(lo[:, 0]).all(), tag
            assert (np.abs(g["region_box_u_lo"][r][reg] - [-d.delta_max, -d.a_max]) <= 4e-15).all(), tag
            assert (np.abs(g["region_box_u_hi"][r][reg] - [d.delta_max, d.a_max]) <= 4e-15).all(), tag
            pr = helpers.oracle_planner_probe(orc, d, p["x0"][q], p["bez_s"][q], p["bez_ey"][q], p["ey_lb"][q], p["ey_ub"][q], g["region_probe_U"][r][reg])
            assert _rel(pr["cost"], g["region_probe_f"][r][reg]) <= 1e-11, (tag, reg, pr["cost"], g["region_probe_f"][r][reg])


def _planner_compare(binding, orc, AB, T, draw=PLANNER_DRAWS[0]):
    from crx import abi
    A, B = AB
    fixture, V, seed, _ = draw
    g = _group(_load(fixture), "draw")
    p = planner_batch(V, seed)
    N, R = 12, V + 1
    idx = g["index"].astype(int)
    rows = np.concatenate([np.arange(b * R, (b + 1) * R) for b in idx])
    d = abi.planner_desc(N, A, B)
    d.opts.tol = T["tol"]
    res = binding.planner_solve(d, *[p[k][rows] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")])
    ok_ref = g["region_success"].reshape(-1)
    st = res["status"]
    # verdicts: the reference's QP is infeasible (HiGHS) <=> status != 0 -> fall-back trajectory, exactly
    np.testing.assert_array_equal(st == 0, ok_ref)
    Xg = g["region_X"].reshape(-1, N + 1, 6)
    dX = np.abs(res["X"] - Xg)
    assert dX[~ok_ref].max() <= 1e-12                                         # closed-form fall-backs (:365-374)
    # the QP weights only s and ey (and the progress term): those, and vx, are sharp; vy, wz, epsi are determined through
    # the dynamics alone (reduced Hessian cond 6.7e6) -- worst of the 129 feasible regions 5.3e-4 at tol 1e-8, 6.7e-6 at 1e-11
    px, pxw = (1e-3, 1e-5) if T["tol"] > 1e-10 else (1e-5, 1e-7)
    assert dX[ok_ref].max() <= px and dX[ok_ref][:, :, [0, 4, 5]].max() <= pxw, (dX[ok_ref].max(), dX[ok_ref][:, :, [0, 4, 5]].max())
    fg = g["region_cert"].reshape(-1, 6)[:, 0]
    assert (_rel(res["cost"][ok_ref], fg[ok_ref]) <= T["f"]).all()
    # selection (a2): direction flag and winning trajectory, from the solver's own X
    sd = abi.select_desc(N, V, LAP)
    n = len(idx)
    sel = orc.select(sd, p["n_veh"][idx], res["X"].reshape(n, R, N + 1, 6), p["obs_s"][idx], p["obs_ey"][idx], p["old_flag"][idx])
    # (two regions whose bounds are not active are the SAME QP: their selection costs tie to 1e-12 in the reference's own numbers and the
    # arg-min is decided by rounding -- seen with five cars, scenario 16: 82.20997584068166 vs ...279.  A flag may differ only inside such a tie.)
    diff = np.nonzero(sel["flag"] != g["direction_flag"])[0]
    for i in diff:
        c = sel["sel_cost"][i]
        assert abs(c[sel["flag"][i]] - c[int(g["direction_flag"][i])]) <= 1e-6 * max(1.0, abs(c[sel["flag"][i]])), (i, c)
    assert len(diff) <= max(1, n // 20), diff
    same = sel["flag"] == g["direction_flag"]
    np.testing.assert_allclose(sel["best_X"][same], g["traj_xcurv"][same], rtol=0, atol=px)
    np.testing.assert_allclose(sel["best_X"][same][:, :, [0, 4, 5]], g["traj_xcurv"][same][:, :, [0, 4, 5]], rtol=0, atol=pxw)
    return res, g


@pytest.mark.parametrize("draw", PLANNER_DRAWS, ids=PLANNER_IDS)
@pytest.mark.parametrize("T", [DEFAULT, TIGHT], ids=["tol1e-8", "tol1e-11"])
def test_oracle_planner_on_reference_built_draw(orc, AB, T, draw):
    res, g = _planner_compare(orc, orc, AB, T, draw)
    ok = g["region_success"].reshape(-1)
    assert ok.sum() >= 0.3 * len(ok) and (~ok).sum() >= 0.2 * len(ok)        # the BASELINE draw: ~41 % infeasible regions


# ---------------------------------------------------------------------------------------------------------------------
# obstacle vehicles of unequal size (control.py:529-535): crx_cbf_solve_dims
# ---------------------------------------------------------------------------------------------------------------------
def dims_batch():
    """The scenarios of tests/golden/cfg2_dims.npz through the PRODUCT's host prep: window test, packing, per-obstacle
    (l_agent + l_obs, w_agent + w_obs) following the kept cars."""
    from crx import hostprep
    g = _group(_load("cfg2_dims.npz"), "draw")
    n, N = len(g["index"]), 12
    j = np.arange(N + 1)
    cars, cd = g["cars"], g["car_dims"]
    obs_s = cars[:, :, 0, None] + 0.1 * j[None, None, :] * cars[:, :, 1, None]
    obs_ey = np.repeat(cars[:, :, 2, None], N + 1, axis=2)
    keep, lap_off = hostprep.cbf_window(g["x0"], obs_s[:, :, 0], LAP)
    dims = np.stack([0.2 + 0.5 * cd[:, :, 0], 0.1 + 0.5 * cd[:, :, 1]], axis=2)          # ego: CarParam() = 0.4 x 0.2
    ps, pe, po, nn, pd = hostprep.pack_obstacles(keep, obs_s, obs_ey, lap_off, 2, dims=dims)
    xt = np.tile(np.array([0.8, 0, 0, 0, 0, 0.0]), (n, 1))
    return g, dict(x0=g["x0"], xt=xt, obs_s=ps, obs_ey=pe, lap_off=po, n_obs=nn, obs_dims=pd)


def _dims_compare(binding, orc, AB, T):
    from crx import abi
    A, B = AB
    g, p = dims_batch()
    d = abi.cbf_desc(12, 2, A, B, alpha=0.8, margin=0.2)
    d.opts.tol = T["tol"]
    res = binding.cbf_solve(d, *[p[k] for k in KEYS], obs_dims=p["obs_dims"])
    plain = binding.cbf_solve(d, *[p[k] for k in KEYS])                                  # the descriptor's single pair: a different problem
    n_cmp = n_diff = 0
    for r in range(len(g["index"])):
        if not (bool(g["certified"][r]) and res["status"][r] == 0):
            continue
        fg = g["cert"][r][0]
        if abs(res["cost"][r] - fg) / max(1.0, abs(fg)) > T["f"] and res["cost"][r] < fg:
            continue
        n_cmp += 1
        scale = max(1.0, abs(fg) * 1e-6)
        Tl = T.get("loose", T)
        assert abs(res["cost"][r] - fg) / max(1.0, abs(fg)) <= T["f"], (r, res["cost"][r], fg)
        assert np.abs(res["X"][r] - g["X"][r]).max() <= Tl["x"] * scale and np.abs(res["U"][r] - g["U"][r]).max() <= Tl["u"] * scale, r
        n_diff += np.abs(plain["X"][r] - g["X"][r]).max() > 1e-4
    assert n_cmp >= 10 and n_diff >= 3, (n_cmp, n_diff)    # ... and the sizes matter: with one common pair the answers differ
    return res


def test_unequal_cars_rows_and_solutions(orc, AB):
    """Rows: cost / CBF rows of the reference's recorded problem (two obstacle cars of different, random sizes) == the oracle's rows
    with per-obstacle dims; solutions: oracle == certified KKT points."""
    from crx import abi
    A, B = AB
    g, p = dims_batch()
    d = abi.cbf_desc(12, 2, A, B, alpha=0.8, margin=0.2)
    seen2 = 0
    for r in range(len(g["index"])):
        n = int(g["n_obs_ref"][r])
        assert int(p["n_obs"][r]) == n, r
        seen2 += n == 2
        pr = helpers.oracle_cbf_probe(orc, d, p["x0"][r], p["xt"][r], p["obs_s"][r], p["obs_ey"][r], p["lap_off"][r], n, g["probe_U"][r],
                                      g["probe_sigma"][r][:n], obs_dims=p["obs_dims"][r])
        assert _rel(pr["cost"], g["probe_f"][r]) <= 1e-12, r
        assert (_rel(pr["cbf"], g["probe_cbf"][r][:n]) <= 1e-9).all(), (r, pr["cbf"], g["probe_cbf"][r][:n])
    assert seen2 >= 8
    for T in (DEFAULT, TIGHT):
        _dims_compare(orc, orc, AB, T)


# ---------------------------------------------------------------------------------------------------------------------
# [r4] four to six vehicles in the window (control.py:524-562 loops over every vehicle): CRX_MAX_OBS = 6
# ---------------------------------------------------------------------------------------------------------------------
def many_batch():
    """The scenarios of tests/golden/cfg2_many.npz (make_draws.py gen_many: 4..5 cars inside the +-2 vx window, problems built by the
    reference's unmodified control.mpccbf) through the PRODUCT's host prep: window test and packing into six obstacle slots."""
    from crx import hostprep
    g = _group(_load("cfg2_many.npz"), "draw")
    n, N = len(g["index"]), 12
    j = np.arange(N + 1)
    cars = np.where(np.isnan(g["cars"]), 0.0, g["cars"])
    there = ~np.isnan(g["cars"][:, :, 0])
    obs_s = cars[:, :, 0, None] + 0.1 * j[None, None, :] * cars[:, :, 1, None]
    obs_ey = np.repeat(cars[:, :, 2, None], N + 1, axis=2)
    keep, lap_off = hostprep.cbf_window(g["x0"], obs_s[:, :, 0], LAP)
    ps, pe, po, nn = hostprep.pack_obstacles(keep & there, obs_s, obs_ey, lap_off, 6)
    xt = np.tile(np.array([0.8, 0, 0, 0, 0, 0.0]), (n, 1))
    return g, dict(x0=g["x0"], xt=xt, obs_s=ps, obs_ey=pe, lap_off=po, n_obs=nn)


def _many_compare(binding, AB, T):
    from crx import abi
    A, B = AB
    g, p = many_batch()
    d = abi.cbf_desc(12, 6, A, B, alpha=0.8, margin=0.2)
    d.opts.tol = T["tol"]
    res = binding.cbf_solve(d, *[p[k] for k in KEYS])
    if T["tol"] >= 1e-9:
        T = dict(T, f=1e-6)     # four or five active degree-6 rows: the barrier perturbation of the cost is ~1e-7 at tol 1e-8 (as on the three-car draw, TOL4)
    n_cmp = 0
    for r in range(len(g["index"])):
        n = int(g["n_obs_ref"][r])
        assert int(p["n_obs"][r]) == n and n >= 4, (r, p["n_obs"][r], n)                # nothing dropped: every car of the window is in the NLP
        assert res["status"][r] == 0 and bool(g["certified"][r]), (r, res["status"][r])
        fg = g["cert"][r][0]
        df = abs(res["cost"][r] - fg) / max(1.0, abs(fg))
        if df > T["f"] and res["cost"][r] < fg:
            continue                                                                     # a better KKT point of the non-convex NLP
        n_cmp += 1
        Tl = T.get("loose", T)
        assert df <= T["f"], (r, res["cost"][r], fg)
        assert np.abs(res["X"][r] - g["X"][r]).max() <= Tl["x"] and np.abs(res["U"][r] - g["U"][r]).max() <= Tl["u"], (r, np.abs(res["X"][r] - g["X"][r]).max())
        ds = np.abs(res["sigma"][r][:n] - g["sigma"][r][:n])
        assert (ds <= 1e-6 * np.maximum(1.0, np.abs(g["sigma"][r][:n]))).all(), (r, ds.max())
    assert n_cmp >= 10, n_cmp
    return res


def test_many_cars_rows_and_solutions(orc, AB):
    """Rows: cost and every CBF row of the reference's recorded problem (4..5 obstacle cars) == the oracle's at the probe point;
    solutions: oracle == the certified KKT points, at both tolerance sets."""
    from crx import abi
    A, B = AB
    g, p = many_batch()
    d = abi.cbf_desc(12, 6, A, B, alpha=0.8, margin=0.2)
    for r in range(len(g["index"])):
        n = int(g["n_obs_ref"][r])
        pr = helpers.oracle_cbf_probe(orc, d, p["x0"][r], p["xt"][r], p["obs_s"][r], p["obs_ey"][r], p["lap_off"][r], n, g["probe_U"][r], g["probe_sigma"][r][:n])
        assert _rel(pr["cost"], g["probe_f"][r]) <= 1e-12, r
        assert (_rel(pr["cbf"][:n], g["probe_cbf"][r][:n]) <= 1e-9).all(), (r, pr["cbf"], g["probe_cbf"][r][:n])
    assert (g["n_obs_ref"] >= 5).sum() >= 3
    for T in (DEFAULT, TIGHT):
        _many_compare(orc, AB, T)


def test_game_loop_lmpc_infeasibility_is_the_references(orc):
    """29-78 % of the learning-MPC QPs of the benched `game` loop end 'infeasible -> relaxed second attempt'.  Is that the
    reference's formulation, or the mirror's regression / safe-set selection drifting?  160 states of that loop (5 lap phases x 32 races) went
    through the reference's OWN estimate_ABC + control.lmpc under the recording stand-in; HiGHS decided the feasibility of every recorded QP
    (tests/golden/tools/make_draws.py game).  The oracle on the reference-built data: converged exactly on the feasible ones, proved infeasible
    exactly on the others -- 81 of 160 -- and the device loop, from the raw state through its own regression, had reached the same 160 verdicts."""
    g, d, args = helpers.game_draw_inputs()
    r = orc.lmpc_solve(d, *args)
    n_feas, n_inf = helpers.check_game_draw(r, g)
    assert n_feas + n_inf == 160 and n_inf >= 60
    by_phase = {int(p): float((g["lp_status"][g["phase"] == p] == 2).mean()) for p in np.unique(g["phase"])}
    assert by_phase[0] == 0.0 and max(by_phase.values()) >= 0.6          # none at the recorded start, up to three quarters mid-lap
