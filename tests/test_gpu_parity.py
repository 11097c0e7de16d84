"""libcrx (HIP, through the C ABI) against the CPU oracle and the golden fixtures.  GPU box only.

Both sides run the same interior-point iteration with different linear algebra (Riccati recursion on the GPU, dense
condensed Cholesky in the oracle), float64 end to end.  Two tolerance sets, the same ones the ORACLE is held to against
the goldens (tests/test_oracle_golden.py):
  DEFAULT  opts.tol = 1e-8 (IPOPT's default, the product default): the barrier perturbation mu ~ 1e-9 is amplified by the
           1e6..1e8 conditioning of the unweighted directions -> cost-weighted states vx, s, ey 1e-5, other states 5e-4,
           inputs 2e-3, cost 1e-7 relative
  TIGHT    opts.tol = 1e-11: states 2e-7 (weighted 1e-7), INPUTS 2e-6, cost 1e-9 relative -- this is the level at which the
           value the reference's controllers return, u_pred[0,:] (control/control.py:607, :473), is pinned
Status and iteration count: identical, problem by problem; `_disagreements` lists every exception and the tests
assert on the list (`_classify` names what is tolerated and why).
"""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

DEFAULT = dict(tol=1e-8, x=5e-4, u=2e-3, f=1e-7, xw=1e-5)
TIGHT = dict(tol=1e-11, x=2e-7, u=2e-6, f=1e-9, xw=1e-7)
TOLS = pytest.mark.parametrize("T", [DEFAULT, TIGHT], ids=["tol1e-8", "tol1e-11"])
XW, XALL, UALL, FREL = DEFAULT["xw"], DEFAULT["x"], DEFAULT["u"], DEFAULT["f"]


def _report(tag, rows):
    """Append disagreement rows to gpurun_out/parity_report.jsonl when CRX_PARITY_REPORT is set (diagnostics)."""
    import json
    import os

    if os.environ.get("CRX_PARITY_REPORT") and rows:
        import conftest
        os.makedirs(os.path.join(conftest.ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(conftest.ROOT, "gpurun_out", "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"tag": tag, "rows": rows}) + "\n")


def _with_tol(d, tol):
    d.opts.tol = tol
    return d


def _disagreements(rg, ro):
    """Every problem whose verdict or iteration count differs between the kernel and the oracle, as printable rows."""
    sg, so, ig, io = rg["status"], ro["status"], rg["iters"], ro["iters"]
    rows = []
    for i in np.nonzero((sg != so) | (ig != io))[0]:
        both = sg[i] == 0 and so[i] == 0
        rows.append(dict(i=int(i), status=(int(sg[i]), int(so[i])), iters=(int(ig[i]), int(io[i])),
                         kkt=(float(rg["kkt"][i]), float(ro["kkt"][i])),
                         dX=float(np.abs(rg["X"][i] - ro["X"][i]).max()) if both else None))
    return rows


def _classify(rows, tol, restored=frozenset()):
    """Sort disagreement rows into named classes.
      tol_edge   both converged, at most 2 iterations apart: the convergence test E <= tol fell on different sides of
                 the threshold at one iterate (the two factorisations differ in the last bits); same point, checked
      restored   the problem went through the restoration phase on either side (crash state: after the slack
                 restoration the cost is 1e8..1e12 and the reduced Hessian sits beyond 1/eps in condition -- e.g. the first
                 post-restoration inertia test already differs between the Riccati pivots and the dense Cholesky).  Only
                 the class of the outcome (converged / not) is comparable there
      tight_stall  tol < 1e-9 only: one side converged (E <= tol, self-certifying), the other stalled at E <= 1e-6 and gave up
                 -- 1e-11 is below what the oracle's condensed Cholesky (N = 20: cond 1e8) can always reach
      verdict    converged on one side only                                   -> never tolerated outside `restored`
      code       both not converged, different status codes                    -> never tolerated outside `restored`
      other      anything else (same status, iteration counts apart)          -> budgeted per test, listed
    """
    out = dict(tol_edge=[], restored=[], verdict=[], code=[], other=[], tight_stall=[])
    for r in rows:
        sg, so = r["status"]
        ig, io = r["iters"]
        if r["i"] in restored:
            out["restored"].append(r)
        elif (sg == 0) != (so == 0):
            conv, oth = (0, 1) if sg == 0 else (1, 0)
            stall = tol < 1e-9 and r["kkt"][conv] <= tol and r["status"][oth] == 1 and r["kkt"][oth] <= 1e-6
            out["tight_stall" if stall else "verdict"].append(r)
        elif sg != so:
            out["code"].append(r)
        elif sg == 0 and abs(ig - io) <= (2 if tol >= 1e-9 else 5):   # at 1e-11 E hovers around the threshold for a few iterations
            out["tol_edge"].append(r)
        else:
            out["other"].append(r)
    return out


def _assert_same_verdicts(tag, rg, ro, tol=1e-8, restored=frozenset(), max_tol_edge=None, max_other=0, max_restored_verdict=0,
                          max_tight_stall=None):
    """Statuses and iteration counts identical problem by problem, up to the classified and budgeted exceptions of
    `_classify`; everything that is tolerated is still listed in the failure message / the diagnostics report."""
    rows = _disagreements(rg, ro)
    _report(tag, rows)
    c = _classify(rows, tol, restored)
    n = len(rg["status"])
    assert not c["verdict"], (tag, "converged on one side only", c["verdict"][:20])
    assert len(c["tight_stall"]) <= (max(1, int(0.05 * n)) if max_tight_stall is None else max_tight_stall), (tag, "stalls at tol = 1e-11", c["tight_stall"][:20])
    assert not c["code"], (tag, "different failure codes", c["code"][:20])
    if max_tol_edge is None:    # at tol = 1e-11 the threshold sits in the rounding noise of E itself
        max_tol_edge = 1 + int((0.15 if tol < 1e-9 else 0.02) * n)
    assert len(c["tol_edge"]) <= max_tol_edge, (tag, "tolerance-edge flips", c["tol_edge"][:20])
    for r in c["tol_edge"]:
        assert r["dX"] is not None and r["dX"] <= XALL, (tag, r)
    assert len(c["other"]) <= max_other, (tag, "unexplained iteration-count disagreements", c["other"][:20])
    bad = [r for r in c["restored"] if (r["status"][0] == 0) != (r["status"][1] == 0)]
    assert len(bad) <= max_restored_verdict, (tag, "restoration-affected problems converged on one side only", bad[:20])
    for r in c["restored"]:
        if r["status"] == (0, 0):
            assert r["dX"] <= XALL, (tag, r)
    return c


@pytest.fixture(scope="module")
def gpu():
    import crx

    return crx.init()


def _cmp(tag, rg, ro, need_same_status=True, T=DEFAULT):
    sg, so = rg["status"], ro["status"]
    if need_same_status:
        assert (sg == so).all(), (tag, np.nonzero(sg != so)[0][:10], sg[sg != so][:10], so[sg != so][:10])
    both = (sg == 0) & (so == 0)
    assert both.sum() > 0
    assert rg["kkt"][both].max() <= T["tol"]
    dX = np.abs(rg["X"][both] - ro["X"][both])
    assert dX[..., [0, 4, 5]].max() <= T["xw"], (tag, dX[..., [0, 4, 5]].max())
    assert dX.max() <= T["x"], (tag, dX.max())
    dU = np.abs(rg["U"][both] - ro["U"][both]).max()
    assert dU <= T["u"], (tag, dU)
    rel = np.abs(rg["cost"][both] - ro["cost"][both]) / np.maximum(1.0, np.abs(ro["cost"][both]))
    assert rel.max() <= T["f"], (tag, rel.max())
    return both


@TOLS
def test_golden_mpccbf(gpu, orc, AB, golden_mpccbf, T):
    """control.mpccbf NLPs recorded from the reference: the kernel against the certified goldens at the oracle's own
    tolerances -- in particular u_pred[0,:], the only thing mpccbf returns (control.py:607)."""
    A, B = AB
    flips = []
    for name in golden_mpccbf.names:
        g = golden_mpccbf.case(name)
        d, args = helpers.mpccbf_inputs(g, A, B)
        _with_tol(d, T["tol"])
        rg = gpu.cbf_solve(d, *args)
        if not bool(g["success"]):
            assert rg["status"][0] != 0, name
            continue
        assert rg["status"][0] == 0, (name, rg["status"], rg["kkt"], rg["iters"])
        assert rg["kkt"][0] <= T["tol"]
        assert abs(rg["cost"][0] - g["cert"][0]) <= T["f"] * max(1.0, abs(g["cert"][0])), name
        np.testing.assert_allclose(rg["X"][0], g["X"], atol=T["x"], err_msg=name)
        np.testing.assert_allclose(rg["X"][0][:, [0, 4, 5]], g["X"][:, [0, 4, 5]], atol=T["xw"], err_msg=name)
        np.testing.assert_allclose(rg["U"][0], g["U"], atol=T["u"], err_msg=name)
        np.testing.assert_allclose(rg["U"][0, 0], g["u_returned"], atol=T["u"], err_msg=name)
        n = int(g["n_obs_in_problem"])
        if n:
            np.testing.assert_allclose(rg["sigma"][0, :n], g["sigma"], atol=1e-6, err_msg=name)
        ro = orc.cbf_solve(d, *args)
        _cmp(name, rg, ro, T=T)
        if rg["iters"][0] != ro["iters"][0]:
            flips.append((name, int(rg["iters"][0]), int(ro["iters"][0])))
    _report("golden_mpccbf tol=%g" % T["tol"], flips)
    assert len(flips) <= 1 and all(abs(a - b) <= 1 for _, a, b in flips), flips   # tolerance-edge flips (see _classify)


@TOLS
def test_golden_planner_and_selection(gpu, orc, AB, golden_planner, T):
    from crx import abi

    A, B = AB
    for name in golden_planner.names:
        g = golden_planner.case(name)
        if not bool(g["overtake_flag"]):
            continue
        d, args = helpers.planner_inputs(g, A, B)
        _with_tol(d, T["tol"])
        rg = gpu.planner_solve(d, *args)
        ro = orc.planner_solve(d, *args)
        assert ((rg["status"] == 0) == g["region_success"]).all(), (name, rg["status"], g["region_success"])
        for reg, ok in enumerate(g["region_success"]):
            tag = "%s/%d" % (name, reg)
            if ok:
                assert abs(rg["cost"][reg] - g["region_cert"][reg, 0]) <= T["f"] * max(1.0, abs(g["region_cert"][reg, 0])), tag
                np.testing.assert_allclose(rg["X"][reg], g["region_X"][reg], atol=T["x"], err_msg=tag)
                np.testing.assert_allclose(rg["X"][reg][:, [0, 4, 5]], g["region_X"][reg][:, [0, 4, 5]], atol=T["xw"], err_msg=tag)
            else:
                np.testing.assert_allclose(rg["X"][reg], g["region_X"][reg], atol=1e-12, err_msg=tag)
                assert np.isinf(rg["cost"][reg])
        _assert_same_verdicts("golden_planner/" + name, rg, ro, tol=T["tol"], max_tol_edge=1)
        if g["region_success"].any():
            _cmp(name, rg, ro, need_same_status=False, T=T)
        N, V = int(g["N"]), g["obs_pred"].shape[0]
        ds = abi.select_desc(N, V, float(g["lap_length"]))
        sel = gpu.select(ds, np.array([V]), rg["X"][None], g["obs_pred"][None, :, 4, :], g["obs_pred"][None, :, 5, :],
                         np.array([int(g["old_flag"])]))
        assert int(sel["flag"][0]) == int(g["direction_flag"]), name
        np.testing.assert_allclose(sel["best_X"][0][:, [4, 5]], g["traj_xcurv"][:, [4, 5]], atol=T["xw"])


@TOLS
def test_golden_mpc_multi_agents(gpu, orc, AB, golden_planner, T):
    """control.mpc_multi_agents NLPs: (u_pred[0,:], x_pred) as returned at control.py:473."""
    A, B = AB
    flips = []
    for name in golden_planner.names:
        g = golden_planner.case(name)
        if not bool(g["overtake_flag"]) or not bool(g["mma_present"]):
            continue
        d, args = helpers.mma_inputs(g, A, B)
        _with_tol(d, T["tol"])
        rg = gpu.cbf_solve(d, *args)
        assert rg["status"][0] == 0, (name, rg["status"], rg["kkt"], rg["iters"])
        assert abs(rg["cost"][0] - g["mma_cert"][0]) <= T["f"] * max(1.0, abs(g["mma_cert"][0])), name
        np.testing.assert_allclose(rg["X"][0], g["mma_X"], atol=T["x"], err_msg=name)
        np.testing.assert_allclose(rg["X"][0][:, [0, 4, 5]], g["mma_X"][:, [0, 4, 5]], atol=T["xw"], err_msg=name)
        np.testing.assert_allclose(rg["U"][0, 0], g["mma_u"], atol=T["u"], err_msg=name)
        np.testing.assert_allclose(rg["X"][0], g["mma_x_pred"], atol=T["x"], err_msg=name)
        ro = orc.cbf_solve(d, *args)
        _cmp(name, rg, ro, T=T)
        if rg["iters"][0] != ro["iters"][0]:
            flips.append((name, int(rg["iters"][0]), int(ro["iters"][0])))
    _report("golden_mma tol=%g" % T["tol"], flips)
    assert len(flips) <= 1 and all(abs(a - b) <= 1 for _, a, b in flips), flips   # tolerance-edge flips (see _classify)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg2_unfiltered", "cfg4", "cfg4_unfiltered"])
@TOLS
def test_synthetic_cbf_batches(gpu, orc, AB, cfg, T):
    """BASELINE configs[1] at its full batch (256, SURVEY 8d's draw and the round-1 filtered one) and configs[3] at 192
    problems.  (1) With the restoration phase switched off (opts.restore_iters = -1) kernel and oracle must agree problem
    by problem in status AND iteration count -- including the crash states, which then end in a failed line search.
    (2) With the default options the problems that restoration does not touch must be bit-for-bit what they were in (1)
    on both sides; the touched ones (crash states) must end in the same class."""
    from crx import abi, synth

    A, B = AB
    if cfg.startswith("cfg2"):
        p = synth.cfg2_mpccbf(256, safe_start=cfg == "cfg2")
        kw = {}
    else:
        p = synth.cfg4_tracking_cbf(192, safe_start=cfg == "cfg4")
        kw = dict(Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    args = (p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], p["n_obs"])

    budgets = {}

    def both_sides(restore_iters):
        d = abi.cbf_desc(p["N"], p["obs_s"].shape[1], A, B, alpha=p["alpha"], margin=p["margin"], **kw)
        _with_tol(d, T["tol"])
        if restore_iters is not None:
            d.opts.restore_iters = restore_iters
        budgets.update(stall=d.opts.stall_iters, restore=d.opts.restore_iters)
        return gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)

    g0, o0 = both_sides(-1)
    # (without any restoration the crash states of the unfiltered draws crawl for 60..200 iterations; over such a run the last bits of
    # two different factorisations add up to a few iterations: ONE such problem per batch is tolerated, same point required)
    _assert_same_verdicts(cfg + " no restoration", g0, o0, tol=T["tol"], max_other=1 if "unfiltered" in cfg else 0)
    g1, o1 = both_sides(None)                # the budgets of the problem class [r6]: (stall_iters, restore_iters) = (50, 25) for N = 12 with one obstacle slot, (100, 50) otherwise
    RI, SI = budgets["restore"], budgets["stall"]
    assert (SI, RI) == ((50, 25) if cfg.startswith("cfg2") else (100, 50))
    touched = set()
    for r0, r1 in ((g0, g1), (o0, o1)):
        dx = np.abs(r0["X"] - r1["X"]).reshape(len(r0["status"]), -1).max(axis=1) > 0
        touched |= set(np.nonzero((r0["status"] != r1["status"]) | (r0["iters"] != r1["iters"]) | dx)[0].tolist())
    frac = len(touched) / len(g0["status"])
    # [r4] with the crash path "touched" also counts the solves that start or restart from the crash point: up to a quarter of the
    # unfiltered three-car draw (a CBF row violated at the zero start is common there), 12 % of the one-car draw
    assert frac <= ((0.25 if cfg.startswith("cfg4") else 0.12) if "unfiltered" in cfg else 0.02), (cfg, frac)
    keep = np.array([i not in touched for i in range(len(g0["status"]))])
    for k in ("X", "U", "status", "iters", "kkt"):
        np.testing.assert_array_equal(g1[k][keep], g0[k][keep], err_msg=k)     # restoration never fires on a healthy problem
    c = _assert_same_verdicts(cfg, g1, o1, tol=T["tol"], restored=frozenset(touched), max_restored_verdict=max(2, len(touched) // 5))
    # restoration turns failed line searches into defined ends: no problem is left at the iteration cap
    assert (g1["status"] == 1).sum() <= (0 if T["tol"] >= 1e-9 else 2), np.bincount(g1["status"], minlength=4)   # (1e-11 is below the noise floor of a crash state of cost 1e8: its line search ends at a feasible point)
    assert g1["iters"].max() <= max(SI + 1 + RI, 1 + 3 * RI), g1["iters"].max()      # stall trigger + restoration budget (a second restoration shares it) | a crash start's budget: < max_iter
    # bounded effort has a price: a crash state that would have crawled to a KKT point in 50..200 iterations now ends as
    # CRX_RESTORED after at most 76 (one-obstacle class) / 151 (the others) iterations: raise opts.stall_iters / restore_iters to trade latency back for convergence
    assert (g1["status"] == 0).sum() >= 0.95 * (g0["status"] == 0).sum()
    # crash states that do converge carry slacks of 1e2..1e6 (cost 1e6..1e10): their trajectories agree to the default set only
    both = _cmp(cfg, g0, o0, need_same_status=False, T=DEFAULT if "unfiltered" in cfg else T)
    assert both.mean() >= (0.85 if "unfiltered" in cfg else 0.95)
    _report(cfg + " restoration-touched", [dict(i=int(i), gpu=(int(g0["status"][i]), int(g0["iters"][i]), int(g1["status"][i]), int(g1["iters"][i])),
                                                cpu=(int(o0["status"][i]), int(o0["iters"][i]), int(o1["status"][i]), int(o1["iters"][i]))) for i in sorted(touched)])


def test_cbf_slack_start_option(gpu, orc, AB):
    """crx_ipm_opts.slack_start (include/crx.h) on the headline draw: 0 = the reference's zero start + closed-form restoration
    (libcrx 0.1.x), 1 = slacks at their provable lower bounds, 2 (default) = the crash path.  Problems whose rows can be met without
    slack are untouched bit for bit; each setting ends more of the crash states at a KKT point, on the kernel and on the oracle
    alike -- the crash path ALL 256 (>= 253) with a shorter tail; every converged trajectory satisfies its CBF
    rows with the reported slacks (both copies of a slack -- state of stage i, input of stage i-1 -- stay equal)."""
    from crx import abi, synth
    A, B = AB
    p = synth.cfg2_mpccbf(256, N=12, seed=2, safe_start=False)
    args = [p[k] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")]
    mk = lambda ss: abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"], opts=abi.default_opts(slack_start=ss))   # noqa: E731
    g = {ss: gpu.cbf_solve(mk(ss), *args) for ss in (0, 1, 2)}
    o = {ss: orc.cbf_solve(mk(ss), *args) for ss in (0, 1, 2)}
    g0 = g[0]
    assert not (g0["status"] == abi.CRX_INFEASIBLE).any()       # x0 is inside its box everywhere: nothing here is PROVED infeasible
    ds = (p["obs_s"][:, 0, 0] + p["lap_off"][:, 0] - p["x0"][:, 4]) / 0.4
    de = (p["obs_ey"][:, 0, 0] - p["x0"][:, 5]) / 0.2
    near = (p["n_obs"] > 0) & (ds ** 6 + de ** 6 < 60.0)                             # inside or next to the safety set at the start
    for ss in (1, 2):
        g1, o1 = g[ss], o[ss]
        touched = (np.abs(g1["X"] - g0["X"]).reshape(256, -1).max(axis=1) > 0) | (g1["iters"] != g0["iters"]) | (g1["status"] != g0["status"])
        assert 4 <= touched.sum() <= 40, int(touched.sum())                           # the crash states of the draw, nothing else
        assert near[touched].all()
        assert (g1["status"] == 0).sum() >= (g0["status"] == 0).sum() + 5 and (o1["status"] == 0).sum() >= (o[0]["status"] == 0).sum() + 5
        ok = g1["status"] == 0
        X, sg = g1["X"][ok], g1["sigma"][ok][:, 0]
        dn = (X[:, :, 4] - p["obs_s"][ok][:, 0]) / 0.4
        dc = (X[:, :, 4] - p["obs_s"][ok][:, 0] - p["lap_off"][ok][:, :1]) / 0.4
        dey = (X[:, :, 5] - p["obs_ey"][ok][:, 0]) / 0.2
        hn, hc = dn ** 6 + dey ** 6 - 1.2 - sg, dc ** 6 + dey ** 6 - 1.2 - sg
        row = np.where((p["n_obs"][ok] > 0)[:, None], hn[:, 1:] - (1 - p["alpha"]) * hc[:, :-1], np.inf)
        assert row.min() >= -1e-6 * max(1.0, 1e-9 * np.abs(hn).max()), (ss, row.min())
    # the crash path: the whole headline batch, kernel and oracle, in fewer iterations than the crawl took
    assert (g[2]["status"] == 0).sum() >= 253 and (o[2]["status"] == 0).sum() >= 253, (np.bincount(g[2]["status"]), np.bincount(o[2]["status"]))
    assert g[2]["iters"].max() <= 45 < g0["iters"].max(), (g[2]["iters"].max(), g0["iters"].max())
    same = (g[2]["status"] == 0) & (o[2]["status"] == 0)
    rel = np.abs(g[2]["cost"][same] - o[2]["cost"][same]) / np.maximum(1.0, np.abs(o[2]["cost"][same]))
    assert (rel <= 1e-6).mean() >= 0.99, rel.max()               # same KKT point on kernel and oracle (a near-tie between two candidates may pick differently)
    # [r4b] slack_start = 3, the EAGER crash start (every violated zero start takes the candidate point; no restart): the whole batch, 30
    # iterations at most (VERDICT r3 item 3's mark, at the price of leaving the reference's local minimum on near-misses: not the default),
    # kernel = oracle; the problems whose zero start violates nothing keep their bits
    g3, o3 = gpu.cbf_solve(mk(3), *args), orc.cbf_solve(mk(3), *args)
    # ([r6] 31, was 30: IPOPT's complete termination test keeps the longest crash start for one more iteration, until its complementarity is below 1e-4)
    assert (g3["status"] == 0).all() and (o3["status"] == 0).all() and g3["iters"].max() <= 31 and o3["iters"].max() <= 31, (g3["iters"].max(), o3["iters"].max())
    assert (g3["iters"] == o3["iters"]).mean() >= 0.98 and (np.abs(g3["cost"] - o3["cost"]) <= 1e-6 * np.maximum(1.0, np.abs(o3["cost"]))).mean() >= 0.99
    same3 = (np.abs(g3["X"] - g[2]["X"]).reshape(256, -1).max(axis=1) == 0) & (g3["iters"] == g[2]["iters"])
    assert 0.85 <= same3.mean() < 1.0, same3.mean()


@pytest.mark.parametrize("N", [12, 20])
def test_synthetic_planner_batch_and_selection(gpu, orc, AB, N):
    from crx import abi, synth

    A, B = AB
    p = synth.cfg3_planner(128, N=N)
    d = abi.planner_desc(N, A, B)
    args = (p["x0"], p["bez_s"], p["bez_ey"], p["ey_lb"], p["ey_ub"])
    rg = gpu.planner_solve(d, *args)
    ro = orc.planner_solve(d, *args)
    # the verdict (converged vs infeasible -> fall-back) must be identical problem by problem: it selects the code path
    # the reference takes (overtake_traj_planner.py:361-374)
    _assert_same_verdicts("cfg3 N=%d" % N, rg, ro)
    _cmp("planner", rg, ro, need_same_status=False)
    fb = rg["status"] != 0
    np.testing.assert_allclose(rg["X"][fb], ro["X"][fb], atol=1e-12)
    assert np.isinf(rg["cost"][fb]).all()
    V, S = p["V"], p["n_scen"]
    ds = abi.select_desc(N, V, p["lap_length"])
    Xs = rg["X"].reshape(S, V + 1, N + 1, 6)
    sg = gpu.select(ds, p["n_veh"], Xs, p["obs_s"], p["obs_ey"], p["old_flag"])
    so = orc.select(ds, p["n_veh"], Xs, p["obs_s"], p["obs_ey"], p["old_flag"])
    assert (sg["flag"] == so["flag"]).all()                      # integer work: bit-exact
    # the reference adds 100 per collision sequentially (:223,:237); the kernel adds 100*count once,
    # which may round the float cost differently in the last place
    np.testing.assert_allclose(sg["sel_cost"], so["sel_cost"], rtol=1e-13)
    np.testing.assert_array_equal(sg["best_X"], so["best_X"])
    # the fused entry point is the same two launches on one stream: identical bits
    pl = gpu.planner_plan(d, ds, *args, p["n_veh"], p["obs_s"], p["obs_ey"], p["old_flag"])
    for k in ("X", "U", "status", "iters"):
        np.testing.assert_array_equal(pl[k], rg[k])
    for k in ("flag", "sel_cost", "best_X"):
        np.testing.assert_array_equal(pl[k], sg[k])


def test_reach_screen_changes_no_verdict(gpu, orc, AB):
    """The reachability screen of the planner QPs (include/crx.h crx_ipm_opts.reach_screen): with it and without it every region gets
    the same verdict, the same trajectory (the fall-back one where the QP has none) and the same selection; the screened regions
    report 0 iterations, all others the iterations they had; it catches nearly all infeasible regions of the BASELINE draw; and
    the kernel screens exactly the regions the oracle screens."""
    import crx
    from crx import abi, synth
    A, B = AB
    N = 12
    p = synth.cfg3_planner(512, N=N, seed=3)
    d = abi.planner_desc(N, A, B)
    args = [p[k] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")]
    on = gpu.planner_solve(d, *args)
    off = gpu.planner_solve(abi.planner_desc(N, A, B, opts=abi.default_opts(reach_screen=0)), *args)   # crx_ipm_opts.reach_screen: per call, no global
    np.testing.assert_array_equal(on["status"] != 0, off["status"] != 0)
    assert (off["status"][off["status"] != 0] == abi.CRX_INFEASIBLE).mean() >= 0.99   # ... proved by the multiplier certificate then (a rare one: CRX_STALLED)
    np.testing.assert_array_equal(on["X"], off["X"])
    np.testing.assert_array_equal(on["U"], off["U"])
    screened = (on["status"] == abi.CRX_INFEASIBLE) & (on["iters"] == 0)
    assert (off["iters"][screened] >= 1).all() and (on["iters"][~screened] == off["iters"][~screened]).all()
    assert np.isinf(on["kkt"][screened]).all() and np.isinf(on["cost"][screened]).all()
    n_bad = int((on["status"] != 0).sum())
    assert 0.3 * len(screened) < n_bad < 0.5 * len(screened) and screened.sum() >= 0.9 * n_bad, (int(screened.sum()), n_bad)
    ro = orc.planner_solve(d, *args)
    np.testing.assert_array_equal(ro["status"], on["status"])
    np.testing.assert_array_equal((ro["status"] == abi.CRX_INFEASIBLE) & (ro["iters"] == 0), screened)


@pytest.mark.parametrize("N,zero_obs", [(12, False), (20, False), (10, True)])
def test_convex_rows_two_methods_one_solution(gpu, orc, AB, N, zero_obs):
    """[r6] crx_ipm_opts.qp_method (include/crx.h): the all-linear problems -- planner region QPs, MPC-CBF NLPs without an obstacle slot -- are strictly
    convex QPs with ONE solution, so Mehrotra's predictor-corrector (0, shipped) and IPOPT's filter line search (1, libcrx <= 0.3) must agree on
    every verdict and on the solution to the tolerance both are solved to; the predictor-corrector takes fewer iterations (the point of it), kernel
    and oracle run the same arithmetic (same statuses, iteration counts equal on nearly all), and the infeasibility proofs are untouched."""
    from crx import abi, synth

    A, B = AB
    if zero_obs:      # the tracking problems of test_zero_obstacle_nlps_incl_infeasible: a narrow track, a third of them infeasible
        Bn = 512
        rng = np.random.default_rng(78)
        x0 = np.zeros((Bn, 6))
        x0[:, 0] = rng.uniform(0.8, 2.0, Bn); x0[:, 1] = rng.uniform(-0.2, 0.2, Bn); x0[:, 3] = rng.uniform(-1.2, 1.2, Bn)
        x0[:, 4] = rng.uniform(0, 5, Bn); x0[:, 5] = rng.uniform(-0.14, 0.14, Bn)
        xt = np.zeros((Bn, 6)); xt[:, 0] = 1.5
        z = np.zeros((Bn, 0, N + 1))
        mk = lambda qm: abi.cbf_desc(N, 0, A, B, ey_max=0.15, v_min=0.5, v_max=2.2, opts=abi.default_opts(qp_method=qm))   # noqa: E731
        args = [x0, xt, z, z, np.zeros((Bn, 0)), np.zeros(Bn, dtype=np.int32)]
        solve_g, solve_o = gpu.cbf_solve, orc.cbf_solve
    else:
        p = synth.cfg3_planner(512, N=N, seed=7)
        mk = lambda qm: abi.planner_desc(N, A, B, opts=abi.default_opts(qp_method=qm))   # noqa: E731
        args = [p[k] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")]
        solve_g, solve_o = gpu.planner_solve, orc.planner_solve
    g0, g1, o0 = solve_g(mk(0), *args), solve_g(mk(1), *args), solve_o(mk(0), *args)
    np.testing.assert_array_equal(g0["status"], g1["status"])
    np.testing.assert_array_equal(g0["status"], o0["status"])
    ok = g0["status"] == 0
    assert ok.sum() >= 0.4 * len(ok)
    # one solution: both methods at tol 1e-8 sit within the conditioning of the QP (lambda_min 1e-6: the uncosted states are only as sharp as 3e-4)
    assert np.abs(g0["X"][ok] - g1["X"][ok])[..., [0, 4, 5]].max() <= 1e-4            # (the cost-weighted states; measured 4e-6)
    assert (np.abs(g0["cost"][ok] - g1["cost"][ok]) / np.maximum(1.0, np.abs(g1["cost"][ok]))).max() <= 1e-8
    assert np.abs(g0["X"][ok] - o0["X"][ok])[..., [0, 4, 5]].max() <= 1e-5            # kernel = oracle, same algorithm
    assert (g0["iters"][ok] == o0["iters"][ok]).mean() >= 0.97
    assert g0["iters"][ok].mean() <= 0.8 * g1["iters"][ok].mean(), (g0["iters"][ok].mean(), g1["iters"][ok].mean())
    assert g0["kkt"][ok].max() <= 1e-8 and g1["kkt"][ok].max() <= 1e-8
    # proofs: the same regions are screened out / certified infeasible, after no more iterations than before
    bad = g0["status"] == abi.CRX_INFEASIBLE
    assert (g0["iters"][bad] <= g1["iters"][bad]).mean() >= 0.95


def test_golden_lmpc(gpu, orc, golden_racing_game):
    """Learning-MPC QPs recorded from the reference's LMPC lap: kernel (block Cholesky in LDS) vs the
    oracle (full KKT LU) vs the certified goldens."""
    g = golden_racing_game
    d, args = helpers.lmpc_inputs(g)
    rg, ro = gpu.lmpc_solve(d, *args), orc.lmpc_solve(d, *args)
    ok = g["lmpc_success"]
    assert (rg["status"][ok] == 0).all(), rg["status"][ok]
    assert rg["kkt"][ok].max() <= 1e-8
    assert np.abs(rg["X"][ok] - g["lmpc/X"][ok]).max() <= 5e-6
    assert np.abs(rg["U"][ok] - g["lmpc/U"][ok]).max() <= 5e-6
    assert np.abs(rg["X"][ok] - ro["X"][ok]).max() <= 5e-6
    rel = np.abs(rg["cost"][ok] - ro["cost"][ok]) / np.abs(ro["cost"][ok])
    assert rel.max() <= 1e-8
    # same verdict and iteration count on every recorded instance (incl. the relaxed second attempt on the infeasible ones)
    _assert_same_verdicts("lmpc recorded", rg, ro)
    # batch entries are independent
    r1 = gpu.lmpc_solve(d, *[a[3:9] for a in args])
    np.testing.assert_array_equal(r1["X"], rg["X"][3:9])


def test_lmpc_reach_screen_skips_only_the_first_attempt(gpu, orc, golden_racing_game):
    """The terminal-set reachability screen of the learning-MPC QP (include/crx.h crx_ipm_opts.reach_screen): with it and without it
    every recorded QP ends with the same status, plan, inputs and hull weights; the screened ones (first attempt skipped) report
    fewer iterations, nothing else changes; several of the recorded infeasible QPs are caught; kernel and oracle agree on which."""
    import crx
    d, args = helpers.lmpc_inputs(golden_racing_game)
    import copy
    on = gpu.lmpc_solve(d, *args)
    d_off = copy.deepcopy(d)
    d_off.opts.reach_screen = 0
    off = gpu.lmpc_solve(d_off, *args)
    for k in ("status", "X", "U", "lam", "cost"):
        np.testing.assert_array_equal(on[k], off[k])
    fewer = on["iters"] < off["iters"]
    assert (on["iters"] <= off["iters"]).all() and fewer.sum() >= 4 and (on["status"][fewer] == 2).all()
    ro = orc.lmpc_solve(d, *args)
    ro_off = orc.lmpc_solve(d_off, *args)
    np.testing.assert_array_equal(ro["iters"] < ro_off["iters"], fewer)      # the same QPs are screened on both sides
    np.testing.assert_array_equal(ro["status"], on["status"])


def test_lmpc_noise_floor_qps(gpu, orc):
    """Eight learning-MPC QPs captured from the batched closed loop (tools/lmpc_stragglers.py) whose local model is unstable
    (|A| entries of 50, as in the reference's own recorded models): the free response reaches 1e7 and the KKT error of the
    relaxed second attempt cannot go below ~1e-6.  They used to run to max_iter (218 iterations in total, holding a
    1024-race launch for 6 ms); the first attempt now ends with the infeasibility proof and the second with the
    stagnation rule (25 iterations with mu < 1e-6), on the device and in the oracle alike."""
    import os

    import conftest
    from crx import abi
    z = np.load(os.path.join(conftest.ROOT, "tests", "golden", "lmpc_noise_floor.npz"))
    d = abi.lmpc_desc(12, 44)
    args = [z[k] for k in ("x0", "u_old", "A", "B", "C", "ss", "qfun", "n_ss")]
    rg, ro = gpu.lmpc_solve(d, *args), orc.lmpc_solve(d, *args)
    assert (rg["status"] != 0).all() and (ro["status"] != 0).all()
    assert rg["iters"].max() <= 80 and ro["iters"].max() <= 80, (rg["iters"], ro["iters"])
    np.testing.assert_array_equal(rg["status"], ro["status"])
    assert np.abs(rg["iters"] - ro["iters"]).max() <= 2, (rg["iters"], ro["iters"])   # the first attempt's proof may come one iteration apart
    # the "last iterate" the reference would consume (control.py:708-722): same plan from both
    assert np.abs(rg["U"][:, 0] - ro["U"][:, 0]).max() <= 1e-4


def test_device_prep(gpu, golden_planner):
    """crx_planner_prep (Bezier references + ey bounds on the device) against the host prep of the mirror
    (planner_helper / hostprep, themselves pinned to the reference at 1e-13 by test_host_mirror) and
    against the Bezier polylines the reference itself computed in the recorded scenarios."""
    import os

    import conftest
    from crx import abi, hostprep, synth
    from planning import planner_helper as ph

    p = synth.cfg3_planner(256, N=12)
    w = p["raw"]
    d = abi.prep_desc(12, 3, len(w["opt_s"]), w["track_width"], w["lap_length"])
    r = gpu.planner_prep(d, w["x"], w["x"], w["n_veh"], w["veh_info"], w["max_dv"], w["obs_s"], w["obs_ey"],
                         w["opt_s"], w["opt_ey"])
    # pow() in the Bernstein weights may differ from libm's by an ulp
    np.testing.assert_allclose(r["bez_s"], p["bez_s"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(r["bez_ey"], p["bez_ey"], rtol=0, atol=1e-14)
    np.testing.assert_array_equal(r["ey_lb"], p["ey_lb"])
    np.testing.assert_array_equal(r["ey_ub"], p["ey_ub"])
    np.testing.assert_array_equal(r["x0"], p["x0"])
    # ragged vehicle counts: regions 0..n_veh match the host prep of a scenario with only those vehicles
    nv = (np.arange(256) % 4).astype(np.int32)
    rr = gpu.planner_prep(d, w["x"], w["x"], nv, w["veh_info"], w["max_dv"], w["obs_s"], w["obs_ey"], w["opt_s"], w["opt_ey"])
    lb, _ = hostprep.planner_ey_bounds(w["x"], w["obs_s"], w["obs_ey"], nv, w["track_width"], w["lap_length"], 12)
    for i in range(0, 256, 7):
        n = int(nv[i])
        if n == 0:
            continue
        cp = ph.bezier_control_points(n, w["veh_info"][i, :n], w["max_dv"][i], 0.5, w["track_width"], w["lap_length"], 0.2, w["opt"], w["x"][i])
        np.testing.assert_allclose(rr["bez_ey"].reshape(256, 4, 13)[i, :n + 1], ph.bezier_polylines(cp, 12)[:, :, 1], rtol=0, atol=1e-14)
        np.testing.assert_array_equal(rr["ey_lb"].reshape(256, 4, 12)[i, :n + 1], lb[i, :n + 1])
    # the reference's own polylines
    opt = np.genfromtxt(os.path.join(conftest.ROOT, "data/optimal_traj/xcurv_l_shape.csv"), delimiter=",")
    for name in golden_planner.names:
        c = golden_planner.case(name)
        if not bool(c["overtake_flag"]):
            continue
        N = int(c["N"])
        names = [str(x) for x in c["veh_names"]]
        interest = [n for n, f in zip(names, c["veh_is_interest"]) if f]
        xc = {n: c["veh_xcurv"][i] for i, n in enumerate(names)}
        pred = {str(n): c["obs_pred"][i] for i, n in enumerate(c["sorted_vehicles"])}
        V = len(interest)
        vi = np.array([[xc[n][4], pred[n][5].max(), pred[n][5].min()] for n in interest])[None]
        mdv = np.array([max(abs(c["x_raw"][0] - xc[n][0]) for n in interest)])
        dg = abi.prep_desc(N, V, opt.shape[0], float(c["width"]), float(c["lap_length"]))
        g = gpu.planner_prep(dg, c["x_wrapped"][None], c["x_raw"][None], np.array([V], np.int32), vi, mdv,
                             c["obs_pred"][None, :, 4, :], c["obs_pred"][None, :, 5, :], opt[:, 4].copy(), opt[:, 5].copy())
        np.testing.assert_allclose(g["bez_s"], c["bezier_xcurvs"][:, :, 0], rtol=0, atol=1e-13, err_msg=name)
        np.testing.assert_allclose(g["bez_ey"], c["bezier_xcurvs"][:, :, 1], rtol=0, atol=1e-13, err_msg=name)
        dd, args = helpers.planner_inputs(c, *[np.eye(6), np.zeros((6, 2))])
        np.testing.assert_array_equal(g["ey_lb"], args[3])
        np.testing.assert_array_equal(g["x0"], args[0])


def test_plant_step(gpu, orc):
    """crx_plant_step (one thread per vehicle, 100 Euler sub-steps) vs the CPU restatement (pinned to the
    reference at 0 ulp by test_oracle_golden) over a lap's worth of states, and in a 60-step closed loop."""
    import os

    import conftest
    from control import control
    from crx import abi
    from utils import racing_env

    spec = np.genfromtxt(os.path.join(conftest.ROOT, "data/track_layout/l_shape.csv"), delimiter=",")
    track = racing_env.ClosedTrack(spec, track_width=1.0)
    tab = track.point_and_tangent
    d = abi.plant_desc(tab.shape[0], track.lap_length)
    rng = np.random.default_rng(11)
    B = 4096
    xc = np.stack([rng.uniform(0.3, 2.0, B), rng.normal(0, 0.05, B), rng.normal(0, 0.3, B), rng.uniform(-0.2, 0.2, B),
                   rng.uniform(-1.0, 2.2 * track.lap_length, B), rng.uniform(-0.8, 0.8, B)], axis=1)
    xg = np.stack([xc[:, 0], xc[:, 1], xc[:, 2], rng.uniform(-3, 3, B), rng.uniform(-5, 5, B), rng.uniform(-5, 5, B)], axis=1)
    u = np.stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-1, 1, B)], axis=1)
    rg, ro = gpu.plant_step(d, tab, xg, xc, u), orc.plant_step(d, tab, xg, xc, u)
    # device sin/cos/atan differ from glibc's in the last place; 100 sub-steps accumulate ~1e-14
    np.testing.assert_allclose(rg["xcurv"], ro["xcurv"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(rg["xglob"], ro["xglob"], rtol=0, atol=1e-12)
    # closed loop under PID, 60 steps (crosses two corners)
    g = (np.zeros((1, 6)), np.zeros((1, 6)))
    o = (np.zeros((1, 6)), np.zeros((1, 6)))
    for k in range(60):
        ug = control.pid(g[1][0], np.array([0.8, 0, 0, 0, 0, 0.0]))[None]
        uo = control.pid(o[1][0], np.array([0.8, 0, 0, 0, 0, 0.0]))[None]
        r = gpu.plant_step(d, tab, g[0], g[1], ug); g = (r["xglob"], r["xcurv"])
        r = orc.plant_step(d, tab, o[0], o[1], uo); o = (r["xglob"], r["xcurv"])
    np.testing.assert_allclose(g[1], o[1], rtol=0, atol=1e-10)
    np.testing.assert_allclose(g[0], o[0], rtol=0, atol=1e-10)
    # with the reference's bounded process noise (crx_plant_step_noise_dev): the reference's own noisy steps
    # (tests/golden/plant_noise.npz, which tests/test_oracle_golden.py holds the oracle to) and a random batch vs the oracle
    import ctypes as C

    import torch

    from crx import torch_api
    gn = np.load(os.path.join(conftest.GOLDEN, "plant_noise.npz"))
    dev = torch.device("cuda", 0)
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)   # noqa: E731

    def noisy(tabn, dn, xgn, xcn, un, zn):
        n = xgn.shape[0]
        o1, o2, laps = torch.empty((n, 6), dtype=torch.float64, device=dev), torch.empty((n, 6), dtype=torch.float64, device=dev), t(np.zeros(n), torch.int32)
        torch_api.plant_step_wrap_dev(dn, t(tabn), t(xgn), t(xcn), t(un), 2, o1, o2, laps, noise_z=t(zn))
        torch.cuda.synchronize()
        return o1.cpu().numpy(), o2.cpu().numpy()

    dn = abi.plant_desc(gn["table"].shape[0], float(gn["lap_length"]))
    xg1, xc1 = noisy(gn["table"], dn, gn["xglob"], gn["xcurv"], gn["u"], gn["z"])
    np.testing.assert_allclose(xc1, gn["xcurv_next"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(xg1, gn["xglob_next"], rtol=0, atol=1e-12)
    z = rng.normal(size=(B, 3)) * rng.choice([1.0, 6.0, 20.0], size=(B, 1))
    keep = xc[:, 4] <= track.lap_length                       # (the wrap variant folds s back; compare where it does not act)
    xg2, xc2 = noisy(tab, d, xg[keep], xc[keep], u[keep], z[keep])
    xgo, xco = np.zeros_like(xg2), np.zeros_like(xc2)
    fn = orc.lib.crx_oracle_plant_step_noise
    fn.restype = C.c_int
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (tab, xg[keep], xc[keep], u[keep], z[keep])]
    assert fn(C.byref(d), C.c_int(int(keep.sum())), *[v.ctypes.data_as(C.c_void_p) for v in a], xgo.ctypes.data_as(C.c_void_p), xco.ctypes.data_as(C.c_void_p)) == 0
    same_lap = xco[:, 4] <= track.lap_length
    np.testing.assert_allclose(xc2[same_lap], xco[same_lap], rtol=0, atol=1e-12)
    np.testing.assert_allclose(xg2[same_lap], xgo[same_lap], rtol=0, atol=1e-12)
    assert same_lap.sum() >= 1000


def test_no_stale_lds_reads(gpu, AB, golden_racing_game):
    """Every solver kernel with its LDS slice pre-filled with NaN (hidden crx_debug_poison_lds): results must be
    bit-identical to the unpoisoned run.  Catches `0 * (LDS the kernel never wrote)`, which is harmless after any
    kernel left finite data behind and nondeterministic on a fresh device."""
    import crx
    from crx import abi, synth

    A, B = AB
    L = crx.lib()
    p2 = synth.cfg2_mpccbf(96, N=12, n_obs=2)
    n2 = (np.arange(96) % 3).astype(np.int32)                       # 0, 1, 2 obstacles present out of 2 slots
    d2 = abi.cbf_desc(12, 2, A, B, alpha=p2["alpha"], margin=p2["margin"])
    a2 = (p2["x0"], p2["xt"], p2["obs_s"], p2["obs_ey"], p2["lap_off"], n2)
    p4 = synth.cfg4_tracking_cbf(64, N=20)
    n4 = (np.arange(64) % 4).astype(np.int32)
    d4 = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    a4 = (p4["x0"], p4["xt"], p4["obs_s"], p4["obs_ey"], p4["lap_off"], n4)
    p3 = synth.cfg3_planner(32, N=12)
    d3 = abi.planner_desc(12, A, B)
    a3 = (p3["x0"], p3["bez_s"], p3["bez_ey"], p3["ey_lb"], p3["ey_ub"])
    dl, al = helpers.lmpc_inputs(golden_racing_game)                # includes instances that take the second attempt

    def run():
        return (gpu.cbf_solve(d2, *a2), gpu.cbf_solve(d4, *a4), gpu.planner_solve(d3, *a3), gpu.lmpc_solve(dl, *al))

    clean = run()
    L.crx_debug_poison_lds(1)
    try:
        dirty = run()
    finally:
        L.crx_debug_poison_lds(0)
    for rc, rd in zip(clean, dirty):
        for k in ("X", "U", "status", "iters", "kkt"):
            np.testing.assert_array_equal(rc[k], rd[k], err_msg=k)


def test_fuzz_descriptors(gpu, orc, AB, golden_racing_game):
    """Differential fuzz over the descriptor space (horizons 3..24 incl. odd ones, 0..3 obstacle slots with ragged
    counts, CBF degree 2/4/6, alpha, margins, weights, per-stage targets; LMPC with ragged safe-set sizes and
    horizons): kernel vs oracle, same verdict and same trajectory wherever both converge."""
    from crx import abi, synth

    A, B = AB
    rng = np.random.default_rng(2024)
    checked = 0
    iter_flips = []
    for trial in range(14):
        N = int(rng.integers(3, 25))
        V = int(rng.integers(0, 4))
        nb = 48
        if V == 0:
            p = synth.cfg2_mpccbf(nb, N=N, seed=100 + trial, n_obs=1)
            d = abi.cbf_desc(N, 0, A, B, Q=tuple(rng.uniform(0, 30, 6)), R=tuple(rng.uniform(0.05, 1.0, 2)), ey_max=float(rng.uniform(0.7, 1.2)))
            args = (p["x0"], p["xt"], np.zeros((nb, 0, N + 1)), np.zeros((nb, 0, N + 1)), np.zeros((nb, 0)), np.zeros(nb, np.int32))
        else:
            per_stage = bool(rng.integers(0, 2))
            p = synth.cfg4_tracking_cbf(nb, N=N, seed=100 + trial, n_obs=V) if per_stage else synth.cfg2_mpccbf(nb, N=N, seed=100 + trial, n_obs=V)
            d = abi.cbf_desc(N, V, A, B, alpha=float(rng.uniform(0.3, 1.0)), margin=float(rng.uniform(0.05, 0.3)),
                             degree=int(rng.choice([2, 4, 6])), per_stage_target=per_stage,
                             Q=(10.0, 0, 0, float(rng.uniform(1, 8)), 0, float(rng.uniform(10, 60))))
            n = rng.integers(0, V + 1, nb).astype(np.int32)
            args = (p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], n)
        d.opts.restore_iters = -1          # exact status / iteration parity is defined without the restoration phase (see _classify)
        rg, ro = gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)
        c = _assert_same_verdicts("fuzz cbf trial %d N=%d V=%d" % (trial, N, V), rg, ro, max_tol_edge=2, max_other=1)
        iter_flips += [(trial, r) for r in c["other"]]
        both = (rg["status"] == 0) & (ro["status"] == 0)
        if both.any():
            dX = np.abs(rg["X"][both] - ro["X"][both])
            assert dX[..., [0, 4, 5]].max() <= XW and dX.max() <= XALL, (trial, N, V, dX.max())
            checked += int(both.sum())
    assert checked >= 300
    # 14 x 48 problems over the whole descriptor space (degrees 2..6, horizons 3..24): converged-vs-not identical but for
    # the listed cases, iteration counts likewise
    assert len(iter_flips) <= 2, iter_flips
    # planner QPs at every horizon class
    for N in (3, 7, 12, 13, 19, 24):
        p = synth.cfg3_planner(16, N=N, seed=N)
        d = abi.planner_desc(N, A, B)
        args = (p["x0"], p["bez_s"], p["bez_ey"], p["ey_lb"], p["ey_ub"])
        rg, ro = gpu.planner_solve(d, *args), orc.planner_solve(d, *args)
        _assert_same_verdicts("fuzz planner N=%d" % N, rg, ro)
        _cmp("planner N=%d" % N, rg, ro, need_same_status=False)
    # learning-MPC QPs: ragged safe-set sizes (first n points of each recorded hull) and shorter horizons
    g = golden_racing_game
    ok = np.nonzero(g["lmpc_success"])[0][:24]
    # 14 and 16 (the second horizon class of the kernel): last stage model repeated; the last case adds a tracking cost
    # (matrix_Q, zero in the reference's LMPCRacingParam), which the kernel accumulates stage by stage at set-up
    for N, Q in ((12, None), (9, None), (5, None), (14, None), (16, None), (12, (1.0, 0.0, 0.0, 0.5, 0.0, 4.0))):
        M = g["lmpc/ss"].shape[2]
        d = abi.lmpc_desc(N=N, n_ss_max=M) if Q is None else abi.lmpc_desc(N=N, n_ss_max=M, Q=Q)
        n_ss = rng.integers(8, M + 1, len(ok)).astype(np.int32)
        idx = np.minimum(np.arange(N), g["lmpc/A"].shape[1] - 1)
        args = (g["lmpc/x"][ok], g["lmpc/u_old"][ok], g["lmpc/A"][ok][:, idx], g["lmpc/B"][ok][:, idx], g["lmpc/C"][ok][:, idx],
                g["lmpc/ss"][ok], g["lmpc/qfun"][ok], n_ss)
        rg, ro = gpu.lmpc_solve(d, *args), orc.lmpc_solve(d, *args)
        # infeasible instances (status 2 = relaxed second attempt): the iteration at which the FIRST attempt is given up
        # (multiplier divergence / no acceptable step on an infeasible QP) is decided by round-off; the total differs then
        _assert_same_verdicts("fuzz lmpc N=%d" % N, rg, ro, max_other=max(1, len(ok) // 6))
        both = (rg["status"] == ro["status"]) & (ro["status"] != 1)
        assert both.sum() >= (12 if N <= 12 else 4), (N, ro["status"])
        assert np.abs(rg["X"][both] - ro["X"][both]).max() <= 1e-5, N
        assert np.abs(rg["U"][both] - ro["U"][both]).max() <= 1e-5, N


def test_cfg4_non_converged_on_the_gpu(gpu, orc, AB):
    """The configs[3] problems that did not converge at round 4's budgets (tests/golden/cfg4_stopped.npz: taken from the benched batch, replayed
    through the reference, classified by the third solver -- tests/test_draw_fixtures.py::test_cfg4_non_converged_are_classified has the story): at
    round 5's defaults (stall rule at 100 iterations, restore_iters = 50) the kernel converges on the ones the oracle converges on, to the same
    points, but for a handful of crash states whose outcome class is decided by rounding (profiles/r04_stress_cbf.txt: 99.97 % same status over
    16 384)."""
    import test_draw_fixtures as tdf

    A, B = AB
    g, p = tdf.cfg4_stopped_batch()
    d = tdf.cbf_desc("cfg4", A, B, tdf.DEFAULT["tol"])
    assert d.opts.restore_iters == 50
    rg, ro = gpu.cbf_solve(d, *[p[k] for k in tdf.KEYS]), orc.cbf_solve(d, *[p[k] for k in tdf.KEYS])
    same = rg["status"] == ro["status"]
    assert same.mean() >= 0.95, (np.nonzero(~same)[0], rg["status"][~same], ro["status"][~same])
    lines, counts = tdf.classify_cfg4_stopped(rg, g)
    print("\n" + str(counts) + "\n  " + "\n  ".join(lines))
    assert counts["converged"] >= 50, counts
    both = (rg["status"] == 0) & (ro["status"] == 0)
    close = np.abs(rg["cost"][both] - ro["cost"][both]) <= 1e-6 * np.maximum(1.0, np.abs(ro["cost"][both]))
    assert close.mean() >= 0.9, (g["index"][both][~close], rg["cost"][both][~close], ro["cost"][both][~close])
    # (crash states crawl for 50..110 iterations: over such a run the last bits of two different factorisations add up to a few iterations)
    di = np.abs(rg["iters"][both][close] - ro["iters"][both][close])
    assert np.median(di) == 0 and (di <= 1).mean() >= 0.85 and di.max() <= 10, di


@pytest.mark.parametrize("N", [12, 10])
def test_speculating_wave_follows_the_sequential_schedule(gpu, AB, N):
    """[r6] The two-wave instantiation crx_solve_kernel<1, 12, 6, N, SPEC = 1> (opt-in experiment, crx_debug_speculation: a second wave factorises the
    reduced Hessian with the NEXT entry of the inertia-correction schedule while the first tries the current one).  (a) With the second wave working
    (mode 1) every output of every problem equals, bit for bit, what the same kernel computes with that wave left idle (mode 2: the schedule one
    attempt at a time) -- the headline draw with its crash states (convexified retries, delta_w sequences, restarts, restorations), the filtered
    draw, lapped cars, both tolerance sets.  (b) Against the shipped one-wave kernel (another compilation of the same source: last-bit differences in
    how the compiler fuses multiply-adds) statuses and iteration counts are identical and the trajectories agree to 1e-6."""
    import crx
    from crx import abi, synth

    A, B = AB
    L = crx.lib()
    n_ic = 0
    for safe, lapped, tol in ((False, 0.0, 1e-8), (True, 0.0, 1e-8), (False, 0.25, 1e-11)):
        p = synth.cfg2_mpccbf(256, N=N, seed=2 if N == 12 else 7, safe_start=safe, lapped_frac=lapped)
        d = abi.cbf_desc(N, 1, A, B, alpha=p["alpha"], margin=p["margin"])
        d.opts.tol = tol
        args = (p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], p["n_obs"])
        try:
            r0 = gpu.cbf_solve(d, *args)          # the default: the one-wave kernel
            L.crx_debug_speculation(2)
            r2 = gpu.cbf_solve(d, *args)
            L.crx_debug_speculation(1)
            r1 = gpu.cbf_solve(d, *args)
            r1b = gpu.cbf_solve(d, *args)
        finally:
            L.crx_debug_speculation(0)
        for k in ("X", "U", "sigma", "cost", "status", "iters", "kkt"):
            np.testing.assert_array_equal(r1[k], r2[k], err_msg=k)
            np.testing.assert_array_equal(r1[k], r1b[k], err_msg=k)
        same = (r0["status"] == r1["status"]) & (r0["iters"] == r1["iters"])
        assert same.mean() >= (1.0 if tol >= 1e-9 else 0.98), np.nonzero(~same)[0]
        assert np.abs(r0["X"][same] - r1["X"][same]).max() <= 1e-6 * max(1.0, tol / 1e-11 * 1e-3 if tol < 1e-9 else 1.0)
        n_ic += int((r0["iters"] > 25).sum())
    assert n_ic >= 1           # the draws hold the long solves the speculation exists for


@pytest.mark.parametrize("kind,n,seed", [("cfg2", 4096, 1), ("cfg4", 2048, 11)])
def test_stress_kernel_vs_oracle_at_scale(gpu, orc, AB, kind, n, seed):
    """[r6] tools/stress_cbf.py as a test: kernel vs oracle on draws 16x larger than the parity tests hold, other seeds than the benched
    ones.  Same status on >= 99.9 % of the problems; every pair BOTH sides report converged whose trajectories differ by more than 1e-5 is CLASSIFIED by
    the oracle-free certificate of tests/kkt_check.py (a linear program over all rows at IPOPT's complementarity tolerance): both points must be KKT
    points of the reference's NLP -- two local solutions of a non-convex problem, the cheaper side recorded -- and such pairs stay below 0.5 %."""
    import kkt_check
    from crx import abi, synth

    A, B = AB
    if kind == "cfg2":
        p = synth.cfg2_mpccbf(n, seed=seed, safe_start=False)
        d = abi.cbf_desc(12, 1, A, B, alpha=0.8, margin=0.2)
    else:
        p = synth.cfg4_tracking_cbf(n, seed=seed, safe_start=False)
        d = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    args = [p[k] for k in kkt_check.KEYS]
    rg, ro = gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)
    sg, so = rg["status"], ro["status"]
    same = sg == so
    print("\n%s x %d (seed %d): converged gpu %d oracle %d; status differs on %s" % (kind, n, seed, (sg == 0).sum(), (so == 0).sum(),
          [(int(b), int(sg[b]), int(rg["iters"][b]), int(so[b]), int(ro["iters"][b])) for b in np.nonzero(~same)[0]]))
    assert same.mean() >= 0.999, same.mean()
    assert (sg == 0).mean() >= (0.9995 if kind == "cfg2" else 0.997)
    lines, table = kkt_check.classify_pairs(d, p, rg, ro)
    print("pairs with |dX| > 1e-5: %s\n  " % table + "\n  ".join(lines))
    assert table["uncertified"] == 0, table
    assert table["pairs"] <= 0.005 * n, table
    both = (sg == 0) & (so == 0)
    dx = np.abs(rg["X"] - ro["X"]).reshape(n, -1).max(axis=1)
    near = both & (dx <= 1e-5)
    rel = np.abs(rg["cost"] - ro["cost"]) / np.maximum(1.0, np.abs(ro["cost"]))
    assert rel[near].max() <= 1e-6, rel[near].max()
    assert (rg["iters"][near] == ro["iters"][near]).mean() >= 0.97


@pytest.mark.parametrize("kind", ["cfg2", "cfg4"])
def test_converged_means_ipopts_complete_test(gpu, orc, AB, kind):
    """[r6] "Converged" = IPOPT's COMPLETE termination test (the reference runs IPOPT on default options: control.py:593): the scaled error <= tol AND the
    unscaled dual infeasibility <= 1, constraint violation <= 1e-4, complementarity <= 1e-4 (crx_ipm_opts.dual_inf_tol / constr_viol_tol /
    compl_inf_tol).  Every converged problem of the headline batch (configs[1], SURVEY 8d draw, crash states included) and of a configs[3] batch meets
    north_star's 1e-4 on violation and complementarity -- read back through the diagnostics switch crx_debug_kkt_unscaled(mode), which changes nothing
    else --, the oracle reports the same components, and with the three tolerances opened the solver is libcrx 0.3's: scaled error only, complementarity
    of a crash state beyond 1e-4."""
    import crx
    from crx import abi, synth

    A, B = AB
    if kind == "cfg2":
        p = synth.cfg2_mpccbf(256, safe_start=False)
        mk = lambda **kw: abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"], opts=abi.default_opts(**{**abi.cbf_class_budgets(12, 1), **kw}))   # noqa: E731
    else:
        p = synth.cfg4_tracking_cbf(2048, N=20, seed=4, safe_start=False)
        mk = lambda **kw: abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True,   # noqa: E731
                                       opts=abi.default_opts(**{**abi.cbf_class_budgets(20, 3), **kw}))
    d = mk()
    assert (d.opts.dual_inf_tol, d.opts.constr_viol_tol, d.opts.compl_inf_tol) == (1.0, 1e-4, 1e-4)
    args = (p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], p["n_obs"])
    r0 = gpu.cbf_solve(d, *args)
    ok = r0["status"] == 0
    assert ok.mean() >= (1.0 if kind == "cfg2" else 0.995)
    parts = {}
    try:
        for mode, name in ((2, "dual"), (3, "viol"), (4, "compl"), (1, "max")):
            crx.lib().crx_debug_kkt_unscaled(mode)
            orc.lib.crx_oracle_debug_kkt_unscaled(mode)
            r1, ro = gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)
            for k in ("X", "U", "sigma", "cost", "status", "iters"):
                np.testing.assert_array_equal(r0[k], r1[k])                # the switch changes kkt[] of the converged problems and nothing else
            np.testing.assert_array_equal(r1["kkt"][~ok], r0["kkt"][~ok])
            parts[name] = r1["kkt"][ok]
            both = ok & (ro["status"] == 0) & (ro["iters"] == r0["iters"])
            assert both.mean() >= 0.97
            # same quantity on both sides (different linear algebra: the last digits of a 1e-9 residual differ)
            assert (np.abs(r1["kkt"][both] - ro["kkt"][both]) <= 1e-6 + 0.5 * np.maximum(r1["kkt"][both], ro["kkt"][both])).mean() >= 0.95, name
    finally:
        crx.lib().crx_debug_kkt_unscaled(0)
        orc.lib.crx_oracle_debug_kkt_unscaled(0)
    print("\n%s: unscaled dual %.2e  violation %.2e  complementarity %.2e (max over %d converged)" % (kind, parts["dual"].max(), parts["viol"].max(), parts["compl"].max(), ok.sum()))
    assert parts["viol"].max() <= 1e-4 and parts["compl"].max() <= 1e-4 and parts["dual"].max() <= 1.0
    np.testing.assert_array_equal(parts["max"], np.maximum(parts["dual"], np.maximum(parts["viol"], parts["compl"])))
    assert (parts["max"] >= r0["kkt"][ok] * (1 - 1e-9)).all()              # every unscaled term dominates its scaled counterpart
    assert np.median(parts["max"]) <= 1e-8
    r2 = gpu.cbf_solve(d, *args)
    np.testing.assert_array_equal(r2["kkt"], r0["kkt"])                    # the switch is off again
    # libcrx 0.3's test (scaled error alone) = the three tolerances opened: same solver, and what it used to accept
    d3 = mk(dual_inf_tol=1e30, constr_viol_tol=1e30, compl_inf_tol=1e30)
    r3 = gpu.cbf_solve(d3, *args)
    assert (r3["iters"] <= r0["iters"]).all() and (r3["iters"] < r0["iters"]).sum() >= 1
    crx.lib().crx_debug_kkt_unscaled(4)
    try:
        c3 = gpu.cbf_solve(d3, *args)
    finally:
        crx.lib().crx_debug_kkt_unscaled(0)
    assert c3["kkt"][c3["status"] == 0].max() > 1e-4


def test_game_loop_lmpc_verdicts_equal_highs(gpu, orc):
    """The kernel on the learning-MPC QPs the REFERENCE builds at the states of the benched `game` loop (tests/golden/game_draw.npz; see
    tests/test_draw_fixtures.py::test_game_loop_lmpc_infeasibility_is_the_references): converged exactly where HiGHS finds the recorded QP
    feasible, proved infeasible exactly where it does not, the certified solutions reproduced; and kernel = oracle on every instance."""
    g, d, args = helpers.game_draw_inputs()
    rg, ro = gpu.lmpc_solve(d, *args), orc.lmpc_solve(d, *args)
    helpers.check_game_draw(rg, g)
    assert (rg["status"] == ro["status"]).all()
    assert np.abs(rg["X"] - ro["X"]).max() <= 1e-5 and np.abs(rg["U"] - ro["U"]).max() <= 1e-5


@pytest.mark.parametrize("N", [7, 16, 24])
def test_general_horizons_against_oracle(gpu, orc, AB, N):
    """The GENERAL instantiations (run-time horizon: every N other than 10 / 12 / 20; csrc/crx_kernels_gen.hip, the conservative
    translation unit) at batch 256 for 0 .. 3 obstacle slots and the generic 4..6-obstacle instantiation: kernel vs oracle on status,
    iteration count, X, U AND sigma (the A/B builds that computed wrong numbers in round 4 were general instantiations,
    and one of them returned a scrambled X with every other output intact).  restore_iters = -1 first (exact status / iteration
    parity is defined without the restoration phase), then the product defaults (crash path on): same verdict classes."""
    from crx import abi, synth

    A, B = AB
    Bn = 256
    total = 0
    for V in (0, 1, 2, 3, 5):
        if V == 0:
            p = synth.cfg2_mpccbf(Bn, N=N, seed=500 + N, n_obs=1)
            d = abi.cbf_desc(N, 0, A, B)
            args = (p["x0"], p["xt"], np.zeros((Bn, 0, N + 1)), np.zeros((Bn, 0, N + 1)), np.zeros((Bn, 0)), np.zeros(Bn, np.int32))
        else:
            per_stage = V >= 2
            p = synth.cfg4_tracking_cbf(Bn, N=N, seed=500 + N + V, n_obs=V) if per_stage else synth.cfg2_mpccbf(Bn, N=N, seed=500 + N + V, n_obs=V)
            d = abi.cbf_desc(N, V, A, B, alpha=p.get("alpha", 0.6) if not per_stage else 0.6, margin=p.get("margin", 0.15) if not per_stage else 0.15,
                             per_stage_target=per_stage, **({"Q": (10.0, 0, 0, 5.0, 0, 50.0)} if per_stage else {}))
            n = np.random.default_rng(N * 10 + V).integers(0, V + 1, Bn).astype(np.int32)
            n[: Bn // 2] = V                                   # half the batch with every slot in use
            args = (p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], n)
        for restore in (-1, None):
            if restore is not None:
                d.opts.restore_iters = restore
            else:
                d.opts.restore_iters = abi.cbf_desc(N, max(V, 0), A, B).opts.restore_iters
            rg, ro = gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)
            tag = "general N=%d V=%d restore_iters=%d" % (N, V, d.opts.restore_iters)
            if restore == -1:
                c = _assert_same_verdicts(tag, rg, ro, max_tol_edge=6, max_other=2)
            else:
                # the crash path: only the class of the outcome is comparable on the problems that took it (see _classify)
                crashy = frozenset(np.nonzero((rg["status"] != ro["status"]) | (rg["iters"] != ro["iters"]))[0].tolist())
                c = _assert_same_verdicts(tag, rg, ro, restored=crashy, max_restored_verdict=max(2, Bn // 50))
                assert len(crashy) <= Bn // 8, (tag, len(crashy))
            both = (rg["status"] == 0) & (ro["status"] == 0)
            same_it = both & (np.abs(rg["iters"] - ro["iters"]) <= 2)
            assert same_it.sum() >= (0.6 if V else 0.5) * Bn, (tag, int(both.sum()), int(same_it.sum()))
            dX = np.abs(rg["X"][same_it] - ro["X"][same_it])
            assert dX[..., [0, 4, 5]].max() <= XW and dX.max() <= XALL, (tag, dX.max())
            assert np.abs(rg["U"][same_it] - ro["U"][same_it]).max() <= UALL, tag
            if V:
                sg, so = rg["sigma"][same_it], ro["sigma"][same_it]
                assert np.abs(sg - so).max() <= 1e-5 * max(1.0, np.abs(so).max()), (tag, np.abs(sg - so).max())
            rel = np.abs(rg["cost"][same_it] - ro["cost"][same_it]) / np.maximum(1.0, np.abs(ro["cost"][same_it]))
            assert rel.max() <= FREL, (tag, rel.max())
            total += int(same_it.sum())
    assert total >= 1500


def test_path_planner_qps(gpu, orc):
    """crx_path_solve (overtake PATH planner, overtake_path_planner.py:199-318) vs the oracle: random references and
    boxes incl. one-sided / missing bounds, active side rows, infeasible boxes and end points, every horizon."""
    from crx import abi

    rng = np.random.default_rng(77)
    for N in (2, 3, 10, 12, 17, 24):
        Bn = 96
        opt, bez = rng.uniform(-0.4, 0.4, (Bn, N + 1)), rng.uniform(-0.9, 0.9, (Bn, N + 1))
        lb, ub = np.full((Bn, N + 1), -1.0), np.full((Bn, N + 1), 1.0)
        side = rng.integers(0, 5, Bn)
        for b in range(Bn):
            j0 = int(rng.integers(0, N)); j1 = int(rng.integers(j0, N + 1))
            if side[b] == 1: ub[b, j0:j1 + 1] = rng.uniform(-0.6, 0.3)
            if side[b] == 2: lb[b, j0:j1 + 1] = rng.uniform(-0.3, 0.6)
            if side[b] == 3: lb[b, :] = -np.inf
            if side[b] == 4: lb[b, j0] = 0.5; ub[b, j0] = 0.3          # empty box
        e0, eN = rng.uniform(-0.5, 0.5, Bn), rng.uniform(-0.5, 0.5, Bn)
        e0[:4] = 1.3                                                   # end point outside its box
        d = abi.path_desc(N, float(rng.uniform(0.5, 1.0)))
        rg, ro = gpu.path_solve(d, opt, bez, lb, ub, e0, eN), orc.path_solve(d, opt, bez, lb, ub, e0, eN)
        np.testing.assert_array_equal(rg["status"], ro["status"])
        ok = ro["status"] == 0
        assert ok.sum() >= Bn // 3 and (ro["status"] == 2).sum() >= 4
        np.testing.assert_allclose(rg["E"][ok], ro["E"][ok], rtol=0, atol=1e-7)
        np.testing.assert_allclose(rg["cost"][ok], ro["cost"][ok], rtol=1e-9)
        np.testing.assert_array_equal(rg["E"][~ok], ro["E"][~ok])
        assert np.isinf(rg["cost"][~ok]).all()
        assert rg["kkt"][ok].max() <= 1e-8


def test_edge_cases(gpu, orc, AB):
    from crx import abi, synth

    A, B = AB
    # empty batch
    d = abi.cbf_desc(10, 1, A, B)
    r = gpu.cbf_solve(d, np.zeros((0, 6)), np.zeros((0, 6)), np.zeros((0, 1, 11)), np.zeros((0, 1, 11)),
                      np.zeros((0, 1)), np.zeros(0, dtype=np.int32))
    assert r["X"].shape == (0, 11, 6)
    # ragged obstacle counts inside one batch (0..3 present out of n_obs_max = 3), maximum horizon
    p = synth.cfg4_tracking_cbf(64, N=abi.CRX_MAX_N)
    n = np.arange(64, dtype=np.int32) % 4
    d = abi.cbf_desc(p["N"], 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    args = (p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], n)
    rg, ro = gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)
    _cmp("ragged", rg, ro, need_same_status=False)
    assert (rg["sigma"][n == 0] == 0).all()
    # argument errors are call failures, not per-problem statuses
    bad = abi.cbf_desc(abi.CRX_MAX_N + 1, 1, A, B)
    with pytest.raises(RuntimeError):
        gpu.cbf_solve(bad, np.zeros((1, 6)), np.zeros((1, 6)), np.zeros((1, 1, abi.CRX_MAX_N + 2)),
                      np.zeros((1, 1, abi.CRX_MAX_N + 2)), np.zeros((1, 1)), np.ones(1, dtype=np.int32))


def test_properties_at_full_size(gpu, AB):
    """BASELINE sizes, oracle-free: every converged problem satisfies the KKT bound; solving a
    batch twice is bit-identical (no cross-problem interference, no uninitialised LDS)."""
    from crx import abi, synth

    A, B = AB
    p = synth.cfg3_planner(1024, N=12)
    d = abi.planner_desc(12, A, B)
    args = (p["x0"], p["bez_s"], p["bez_ey"], p["ey_lb"], p["ey_ub"])
    r1, r2 = gpu.planner_solve(d, *args), gpu.planner_solve(d, *args)
    for k in ("X", "U", "status", "iters", "kkt"):
        np.testing.assert_array_equal(r1[k], r2[k])
    ok = r1["status"] == 0
    assert r1["kkt"][ok].max() <= 1e-8
    # a permuted batch gives the permuted answer
    perm = np.random.default_rng(0).permutation(len(ok))
    r3 = gpu.planner_solve(d, *[a[perm] for a in args])
    np.testing.assert_array_equal(r3["X"], r1["X"][perm])
    # dynamics hold on every returned (converged) trajectory: x_{k+1} = A x_k + B u_k
    X, U = r1["X"][ok], r1["U"][ok]
    res = X[:, 1:] - (X[:, :-1] @ A.T + U @ B.T)
    assert np.abs(res).max() <= 1e-10
    # bounds hold
    assert np.abs(U[..., 0]).max() <= 0.5 + 1e-7 and np.abs(U[..., 1]).max() <= 1.5 + 1e-7


def test_cfg5_shard_full_size(gpu, orc, AB):
    """BASELINE configs[4], one GPU's shard at full size: 16384 raw overtake scenarios (65536 region QPs) through the
    device-resident chain bench.py times -- crx_planner_prep_dev -> crx_planner_solve_dev -> crx_select_dev
    (crx.pipeline.PlannerSweep; the all-gather is the identity on one rank).  Oracle-free properties on all of it,
    oracle parity on a 512-scenario subsample.  Replaces solve_optimization_problem for the sweep of
    car_racing/tests/overtake_planner_test.py:307-309."""
    import torch
    from crx import abi, pipeline, synth

    A, B = AB
    S, N, V = 16384, 12, 3
    R = V + 1
    raw = synth.cfg3_raw(S, N=N, seed=5)
    dev = torch.device("cuda", 0)
    sw = pipeline.PlannerSweep(raw, A, B, S, dev)
    flag, best, _ = sw.step()
    torch.cuda.synchronize()
    out = {k: getattr(sw.ws, k).cpu().numpy() for k in ("X", "U", "status", "iters", "kkt", "cost")}
    flag, best = flag.cpu().numpy().copy(), best.cpu().numpy().copy()
    assert out["X"].shape == (S * R, N + 1, 6) and flag.shape == (S,)
    ok = out["status"] == 0
    assert 0.3 <= ok.mean() <= 0.9                      # a large share of the region QPs is infeasible by construction
    assert set(np.unique(out["status"])) <= {0, 2}       # planner QPs: converged, or infeasible -> fall-back; nothing undefined
    assert out["kkt"][ok].max() <= 1e-8
    X, U = out["X"][ok], out["U"][ok]
    assert np.abs(X[:, 1:] - (X[:, :-1] @ A.T + U @ B.T)).max() <= 1e-10      # dynamics
    assert np.abs(U[..., 0]).max() <= 0.5 + 1e-7 and np.abs(U[..., 1]).max() <= 1.5 + 1e-7
    assert X[:, 1:, 0].max() <= 5.0 + 1e-7
    assert np.isinf(out["cost"][~ok]).all() and (out["U"][~ok] == 0).all()      # fall-back branch (:365-374)
    # winners are rows of X
    Xs = out["X"].reshape(S, R, N + 1, 6)
    np.testing.assert_array_equal(best, Xs[np.arange(S), flag])
    # bit-identical rerun
    f2, b2, s2 = sw.step()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(f2.cpu().numpy(), flag)
    np.testing.assert_array_equal(b2.cpu().numpy(), best)
    # the winner record carries the status of the winning region's QP
    np.testing.assert_array_equal(s2.cpu().numpy(), out["status"].reshape(S, R)[np.arange(S), flag])
    np.testing.assert_array_equal(sw.ws.iters.cpu().numpy(), out["iters"])
    # permutation invariance: scenarios are independent
    perm = np.random.default_rng(1).permutation(S)
    rawp = {k: (v[perm] if isinstance(v, np.ndarray) and v.ndim and v.shape[0] == S else v) for k, v in raw.items()}
    fp, bp, _ = pipeline.PlannerSweep(rawp, A, B, S, dev).step()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(fp.cpu().numpy(), flag[perm])
    np.testing.assert_array_equal(bp.cpu().numpy(), best[perm])
    # oracle on a subsample (the mirror's host prep is a Python loop per scenario): host prep + oracle QPs + oracle selection
    sub = np.sort(np.random.default_rng(2).choice(S, 512, replace=False))
    from crx import hostprep
    from planning import planner_helper as ph
    bez = np.zeros((len(sub), R, N + 1, 2))
    for i, s in enumerate(sub):
        cp = ph.bezier_control_points(V, raw["veh_info"][s], raw["max_dv"][s], 0.5, raw["track_width"], raw["lap_length"], 0.2, raw["opt"], raw["x"][s])
        bez[i] = ph.bezier_polylines(cp, N)
    lb, ub = hostprep.planner_ey_bounds(raw["x"][sub], raw["obs_s"][sub], raw["obs_ey"][sub], raw["n_veh"][sub], raw["track_width"], raw["lap_length"], N)
    d = abi.planner_desc(N, A, B)
    ro = orc.planner_solve(d, np.repeat(raw["x"][sub], R, axis=0), bez[..., 0].reshape(-1, N + 1), bez[..., 1].reshape(-1, N + 1),
                           lb.reshape(-1, N), ub.reshape(-1))
    idx = (sub[:, None] * R + np.arange(R)[None]).reshape(-1)
    rg = {k: out[k][idx] for k in out}
    _assert_same_verdicts("cfg5 subsample", rg, ro)
    _cmp("cfg5 subsample", rg, ro, need_same_status=False)
    fb = ro["status"] != 0
    np.testing.assert_allclose(rg["X"][fb], ro["X"][fb], atol=1e-12)
    so = orc.select(abi.select_desc(N, V, raw["lap_length"]), raw["n_veh"][sub], ro["X"].reshape(len(sub), R, N + 1, 6),
                    raw["obs_s"][sub], raw["obs_ey"][sub], raw["old_flag"][sub])
    np.testing.assert_array_equal(so["flag"], flag[sub])
    np.testing.assert_allclose(so["best_X"][..., [0, 4, 5]], best[sub][..., [0, 4, 5]], atol=XW)


def test_cfg4_full_size(gpu, orc, AB):
    """BASELINE configs[3] at its full batch: 16384 tracking NLPs (N = 20, 3 obstacles, CBF rows) in one launch.
    Oracle-free properties on all of them, oracle parity (exact verdicts) on a 256-problem subsample."""
    from crx import abi, synth

    A, B = AB
    Bn = 16384
    p = synth.cfg4_tracking_cbf(Bn, N=20, seed=4, safe_start=False)
    d = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    args = (p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], p["n_obs"])
    r1 = gpu.cbf_solve(d, *args)
    ok = r1["status"] == 0
    assert ok.mean() >= 0.999, ok.mean()       # [r6] 99.95 % at the class budgets (stall_iters 100, restore_iters 50); the threshold was 0.90 until round 5
    # SURVEY 8d's draw puts ~7 % of the egos inside or about to enter an obstacle's unsafe set (crash states): those end
    # as CRX_RESTORED / CRX_INFEASIBLE; an undefined end (iteration cap) is practically absent
    assert (r1["status"] == 1).mean() <= 2e-3, np.bincount(r1["status"], minlength=4)
    assert r1["kkt"][ok].max() <= 1e-8
    X, U, sg = r1["X"][ok], r1["U"][ok], r1["sigma"][ok]
    assert np.abs(X[:, 1:] - (X[:, :-1] @ A.T + U @ B.T)).max() <= 1e-10
    assert np.abs(U[..., 0]).max() <= 0.5 + 1e-7 and np.abs(U[..., 1]).max() <= 1.0 + 1e-7
    assert sg.min() >= -1e-7
    assert np.abs(X[..., 5]).max() <= 1.0 + 1e-7 and X[..., 0].min() >= -1e-7
    # CBF rows hold on every converged trajectory: h_{i+1} - (1 - alpha) h_i >= 0 with the slack (control.py:527-558)
    # (the obstacles that passed the controller's window test, :293-309; `diffs` lap-corrected, `diffs_next` not: quirk Q1)
    al, cm = 0.6, 1.15
    de = (X[:, None, :, 5] - p["obs_ey"][ok]) / 0.2
    ds_next = (X[:, None, :, 4] - p["obs_s"][ok]) / 0.4
    ds_cur = (X[:, None, :, 4] - p["obs_s"][ok] - p["lap_off"][ok][:, :, None]) / 0.4
    h_next = ds_next ** 6 + de ** 6 - cm - sg
    h_cur = ds_cur ** 6 + de ** 6 - cm - sg
    row = h_next[:, :, 1:] - (1 - al) * h_cur[:, :, :-1]
    present = np.arange(3)[None, :] < p["n_obs"][ok][:, None]
    assert present.mean() >= 0.8 and (~present).sum() >= 100                 # the window test drops ~10 % of the drawn cars
    viol = np.where(present[:, :, None], row, np.inf)
    assert viol.min() >= -1e-6 * max(1.0, np.abs(h_next[present]).max() * 1e-9), viol.min()
    assert (sg[~present] == 0.0).all()                                        # slacks of absent obstacle slots are reported as 0
    # bit-identical rerun, permutation invariance
    r2 = gpu.cbf_solve(d, *args)
    for k in ("X", "U", "status", "iters", "kkt"):
        np.testing.assert_array_equal(r1[k], r2[k])
    perm = np.random.default_rng(3).permutation(Bn)[:4096]
    r3 = gpu.cbf_solve(d, *[a[perm] for a in args])
    np.testing.assert_array_equal(r3["X"], r1["X"][perm])
    np.testing.assert_array_equal(r3["status"], r1["status"][perm])
    sub = np.sort(np.random.default_rng(4).choice(Bn, 256, replace=False))
    ro = orc.cbf_solve(d, *[a[sub] for a in args])
    rg = {k: r1[k][sub] for k in ("X", "U", "status", "iters", "kkt", "cost")}
    # the subsample with the restoration phase off on both sides: exact verdicts and iteration counts
    d.opts.restore_iters = -1
    rg0, ro0 = gpu.cbf_solve(d, *[a[sub] for a in args]), orc.cbf_solve(d, *[a[sub] for a in args])
    _assert_same_verdicts("cfg4 subsample, no restoration", rg0, ro0)
    touched = frozenset(np.nonzero((rg0["status"] != rg["status"]) | (rg0["iters"] != rg["iters"]) | (ro0["status"] != ro["status"]) | (ro0["iters"] != ro["iters"]))[0].tolist())
    _assert_same_verdicts("cfg4 subsample", rg, ro, restored=touched, max_restored_verdict=max(2, len(touched) // 5))
    _cmp("cfg4 subsample", rg0, ro0, need_same_status=False)


def test_lmpc_prep_device(gpu, orc, golden_racing_game):
    """crx_lmpc_prep (regression + linearisation + safe-set selection on the device, one wave per race) against the oracle,
    which tests/test_oracle_golden.py pins to the reference's own recorded stage models.  Kernel and oracle run the same
    operations in the same order without fused multiply-adds: the regression rows must agree to rounding of the final
    divisions although the normal matrices have condition 3e11; the kinematic rows differ by the device's sin / cos."""
    import os

    import conftest
    from utils import racing_env

    g = golden_racing_game
    track = racing_env.ClosedTrack(np.genfromtxt(os.path.join(conftest.ROOT, "data/track_layout/l_shape.csv"), delimiter=","), track_width=1.0)
    d, ss, us, qf, time_ss, lin_points, lin_input = helpers.lmpc_lap_setup(g, track)
    tab, N, L = track.point_and_tangent, d.N, float(g["lap_length"])
    n = int(g["lmpc_first_uncertified"]) + 1
    # the recorded calls as ONE batch: every call has its own copy of the safe set (extended by the calls before it)
    SS, US, LP, LI, XS = [], [], [], [], []
    for c in range(n):
        SS.append(ss.copy()); US.append(us.copy()); LP.append(lin_points.copy()); LI.append(lin_input.copy()); XS.append(g["lmpc/x"][c])
        X, U = g["lmpc/X"][c], g["lmpc/U"][c]
        lin_points, lin_input = np.concatenate((X[1:], X[-1:]), axis=0), np.vstack((U[1:], U[-1]))
        ss[1, time_ss[1] + c + 1] = g["lmpc/x"][c] + np.array([0, 0, 0, 0, L, 0])
        us[1, time_ss[1] + c + 1] = U[0]
    # plus perturbed copies (other linearisation points, other states -> other neighbour sets and nearest points)
    rng = np.random.default_rng(5)
    for c in range(n, 64):
        k = c % n
        SS.append(SS[k]); US.append(US[k])
        LP.append(LP[k] + rng.normal(0, [0.05, 0.01, 0.05, 0.01, 0.1, 0.02], (N + 1, 6)))
        LI.append(LI[k] + rng.normal(0, 0.02, (N, 2)))
        XS.append(XS[k] + rng.normal(0, [0.05, 0.01, 0.05, 0.01, 0.3, 0.02]))
    Bn = len(SS)
    args = (np.stack(SS), np.stack(US), np.tile(qf[None], (Bn, 1, 1)), np.tile(time_ss[None], (Bn, 1)), np.full(Bn, 2, dtype=np.int32),
            np.stack(XS), np.stack(LP), np.stack(LI), tab)
    rg, ro = gpu.lmpc_prep(d, *args), orc.lmpc_prep(d, *args)
    np.testing.assert_array_equal(rg["status"], ro["status"])
    assert (ro["status"] == 0).all()
    np.testing.assert_array_equal(rg["ss"], ro["ss"])                  # selection: index work, bit-exact
    np.testing.assert_array_equal(rg["qfun"], ro["qfun"])
    scale = np.maximum(1.0, np.abs(ro["A"]).max(axis=(2, 3), keepdims=True))
    assert (np.abs(rg["A"][:, :, :3] - ro["A"][:, :, :3]) / scale).max() <= 1e-12, (np.abs(rg["A"][:, :, :3] - ro["A"][:, :, :3]) / scale).max()
    np.testing.assert_allclose(rg["B"], ro["B"], rtol=0, atol=1e-12 * float(scale.max()))
    np.testing.assert_allclose(rg["C"][:, :, :3], ro["C"][:, :, :3], rtol=0, atol=1e-12 * float(scale.max()))
    np.testing.assert_allclose(rg["A"][:, :, 3:], ro["A"][:, :, 3:], rtol=0, atol=1e-13)      # kinematic rows
    np.testing.assert_allclose(rg["C"][:, :, 3:], ro["C"][:, :, 3:], rtol=0, atol=1e-12)
    # and, through the oracle's pin, the reference's own models: predictions at the query points
    for c in range(n):
        Ag, Bg, Cg = g["lmpc/A"][c], g["lmpc/B"][c], g["lmpc/C"][c]
        pred = np.einsum("nij,nj->ni", rg["A"][c], LP[c][:N]) + np.einsum("nij,nj->ni", rg["B"][c], LI[c]) + rg["C"][c]
        pred_g = np.einsum("nij,nj->ni", Ag, LP[c][:N]) + np.einsum("nij,nj->ni", Bg, LI[c]) + Cg
        np.testing.assert_allclose(pred, pred_g, atol=1e-5)
        np.testing.assert_array_equal(rg["ss"][c], g["lmpc/ss"][c])
    # singular stages (no stored sample near the linearisation point; the reference's cvxopt raises): status 1 and the three
    # regression rows of the stage are left as the caller passed them -- through the HOST entry point too (the
    # staging copy used to hand back what an earlier call had left there)
    LPs = np.stack(LP).copy()
    LPs[::3, 4, 0] += 400.0                                            # stage 4 of every third race: vx far off the data
    a3 = args[:6] + (LPs, args[7], tab)
    seed = (rng.normal(size=(Bn, N, 6, 6)), rng.normal(size=(Bn, N, 6, 2)), rng.normal(size=(Bn, N, 6)))
    sg_, so_ = gpu.lmpc_prep(d, *a3, seed=seed), orc.lmpc_prep(d, *a3, seed=seed)
    np.testing.assert_array_equal(sg_["status"], so_["status"])
    assert (so_["status"][::3] == 1).all() and (so_["status"][1::3] == 0).all()
    for k, sd_ in zip("ABC", seed):
        np.testing.assert_array_equal(sg_[k][::3, 4, :3], sd_[::3, 4, :3], err_msg="kernel touched the rows of a singular stage: " + k)
        np.testing.assert_array_equal(so_[k][::3, 4, :3], sd_[::3, 4, :3], err_msg="oracle touched the rows of a singular stage: " + k)
        np.testing.assert_allclose(sg_[k], so_[k], rtol=0, atol=1e-12 * float(scale.max()), err_msg=k)
    # from_plan: the shift of the previous plan done inside the kernel
    Xp, Up = np.stack([g["lmpc/X"][c % n] for c in range(Bn)]), np.stack([g["lmpc/U"][c % n] for c in range(Bn)])
    a2 = args[:6] + (Xp, Up, tab)
    rp = gpu.lmpc_prep(d, *a2, from_plan=True)
    rq = gpu.lmpc_prep(d, *(args[:6] + (np.concatenate((Xp[:, 1:], Xp[:, -1:]), axis=1), np.concatenate((Up[:, 1:], Up[:, -1:]), axis=1), tab)))
    for k in ("A", "B", "C"):
        np.testing.assert_array_equal(rp[k], rq[k])


@pytest.mark.parametrize("V", [3, 6])
def test_scene_device(gpu, orc, AB, V):
    """crx_planner_scene (interest test, the reference's partial ey sort, veh_infos in iteration order, max_delta_v, sorted
    predictions; one wave per scenario) against the oracle, which tests/test_host_mirror.py pins to the reference's recorded
    decisions: integer and copy work, bit-exact.  Then the whole device chain from raw vehicles to winners -- scene -> prep
    -> region QPs -> selection -- against the same chain with the scene stage done by the oracle."""
    from crx import abi

    A, B = AB
    L, N, VA, S = 19.22957795362994, 12, (6 if V == 3 else 8), 4096       # V = 6 = CRX_MAX_VEH [r5]: up to eight vehicles around, six slots
    ego, n_all, veh, ps, pe = helpers.random_scenes(S, VA, N, L, seed=11)
    d = abi.scene_desc(N, VA, V, L)
    rg, ro = gpu.planner_scene(d, ego, n_all, veh, ps, pe), orc.planner_scene(d, ego, n_all, veh, ps, pe)
    for k in ("n_veh", "overflow", "order", "veh_info", "max_dv", "obs_s", "obs_ey"):
        np.testing.assert_array_equal(rg[k], ro[k], err_msg=k)
    assert (rg["n_veh"] == 0).sum() >= 10 and (rg["overflow"] > 0).sum() >= (10 if V == 3 else 1) and (rg["n_veh"] == V).sum() >= (100 if V == 3 else 3)
    assert V == 3 or (rg["n_veh"] > 3).sum() >= 100
    # chain on the scenes the planner would run on (>= 1 vehicle of interest, end point inside the optimal-trajectory table)
    import os

    import conftest
    opt = np.genfromtxt(os.path.join(conftest.ROOT, "data/optimal_traj/xcurv_l_shape.csv"), delimiter=",")
    xw = ego.copy()
    xw[:, 4] = np.where(xw[:, 4] > L, xw[:, 4] - L, xw[:, 4])                      # the wrapped copy the planner is handed (base.py:460-462)
    ok = (rg["n_veh"] > 0) & (xw[:, 4] + 0.5 * rg["max_dv"] + 4.0 < opt[-1, 4] - 0.1)
    idx = np.nonzero(ok)[0][:512]
    dp = abi.prep_desc(N, V, opt.shape[0], 1.0, L)
    pr = gpu.planner_prep(dp, xw[idx], ego[idx], rg["n_veh"][idx], rg["veh_info"][idx], rg["max_dv"][idx], rg["obs_s"][idx], rg["obs_ey"][idx],
                          np.ascontiguousarray(opt[:, 4]), np.ascontiguousarray(opt[:, 5]))
    dq, ds = abi.planner_desc(N, A, B), abi.select_desc(N, V, L)
    old = np.full(len(idx), -1, dtype=np.int32)
    pl = gpu.planner_plan(dq, ds, pr["x0"], pr["bez_s"], pr["bez_ey"], pr["ey_lb"], pr["ey_ub"], rg["n_veh"][idx], rg["obs_s"][idx], rg["obs_ey"][idx], old)
    assert (pl["flag"] <= rg["n_veh"][idx]).all() and (pl["flag"] >= 0).all()        # a region of THIS scenario wins
    qo = orc.planner_solve(dq, pr["x0"], pr["bez_s"], pr["bez_ey"], pr["ey_lb"], pr["ey_ub"])
    so = orc.select(ds, ro["n_veh"][idx], qo["X"].reshape(len(idx), V + 1, N + 1, 6), ro["obs_s"][idx], ro["obs_ey"][idx], old)
    # the selection cost carries -10 (s_N - s_0) of the QP solutions, which agree to ~1e-6 between kernel and oracle: regions
    # that tie closer than that may swap; everywhere else the same region wins
    diff = np.nonzero(pl["flag"] != so["flag"])[0]
    assert len(diff) <= (3 if V == 3 else 12), len(diff)       # (more regions: more pairs of identical QPs that tie)
    for i in diff:
        assert abs(pl["sel_cost"][i, pl["flag"][i]] - pl["sel_cost"][i, so["flag"][i]]) <= 1e-4, (i, pl["sel_cost"][i], so["sel_cost"][i])


def test_track_prep_device(gpu):
    """crx_track_prep_dev (per-stage targets + obstacle window / packing of the tracking NLP, on the device) against the host
    prep of the mirror (crx.hostprep tracking_targets / cbf_window / pack_obstacles, the functions the golden mpc_multi_agents
    tests feed the solver through, tests/helpers.py mma_inputs): same arithmetic: window / packing / offsets bit-exact, interpolated targets to 1e-13 (the kernel contracts a*b+c)."""
    import torch
    from crx import hostprep, torch_api

    rng = np.random.default_rng(31)
    Bn, N, V, L = 512, 10, 3, 19.22957795362994
    x = np.zeros((Bn, 6)); x[:, 0] = rng.uniform(0.5, 1.6, Bn); x[:, 4] = rng.uniform(0, L, Bn); x[:, 5] = rng.uniform(-0.5, 0.5, Bn)
    n_veh = rng.integers(0, V + 1, Bn).astype(np.int32)
    vo = rng.uniform(0.3, 1.2, (Bn, V))
    so = x[:, 4, None] + rng.uniform(-3.0, 4.0, (Bn, V))
    so[::7] += L                                                       # predictions one lap ahead of the ego: lap offsets
    j = np.arange(N + 1)
    obs_s = so[:, :, None] + 0.1 * j * vo[:, :, None]
    obs_ey = rng.uniform(-0.7, 0.7, (Bn, V, 1)) + np.zeros((1, 1, N + 1))
    traj = np.zeros((Bn, N + 1, 6))
    traj[:, :, 4] = x[:, 4, None] + np.cumsum(rng.uniform(0.05, 0.2, (Bn, N + 1)), axis=1)
    traj[:, :, 5] = rng.uniform(-0.6, 0.6, (Bn, N + 1))
    dev = torch.device("cuda", 0)
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
    xt = torch.empty((Bn, N + 1, 6), dtype=torch.float64, device=dev)
    os_, oe_ = torch.empty((Bn, V, N + 1), dtype=torch.float64, device=dev), torch.empty((Bn, V, N + 1), dtype=torch.float64, device=dev)
    lo, no = torch.empty((Bn, V), dtype=torch.float64, device=dev), torch.empty((Bn,), dtype=torch.int32, device=dev)
    torch_api.track_prep_dev(N, V, L, t(x), t(n_veh, torch.int32), t(obs_s), t(obs_ey), t(traj), xt, os_, oe_, lo, no)
    torch.cuda.synchronize()
    for b in range(Bn):
        np.testing.assert_allclose(xt[b].cpu().numpy(), hostprep.tracking_targets(x[b], traj[b], N), rtol=0, atol=1e-13)   # fma in the interpolation (slope * dx cancels)
        nv = int(n_veh[b])
        keep, off = hostprep.cbf_window(x[b:b + 1], obs_s[b:b + 1, :nv, 0], L)
        ps, pe, po, n = hostprep.pack_obstacles(keep, obs_s[b:b + 1, :nv], obs_ey[b:b + 1, :nv], off, V) if nv else (
            np.zeros((1, V, N + 1)), np.zeros((1, V, N + 1)), np.zeros((1, V)), np.zeros(1, dtype=np.int32))
        assert int(no[b]) == int(n[0]), b
        np.testing.assert_array_equal(os_[b].cpu().numpy(), ps[0]); np.testing.assert_array_equal(oe_[b].cpu().numpy(), pe[0])
        np.testing.assert_array_equal(lo[b].cpu().numpy(), po[0])
    assert (no.cpu().numpy() > 0).sum() > 100 and (lo.cpu().numpy() != 0).sum() > 10


def test_masked_launches(gpu, golden_racing_game, AB):
    """crx_*_masked_dev: problems with active == 0 are left alone (status CRX_SKIPPED, outputs untouched), the others are
    solved exactly as by the unmasked entry points (bit-identical)."""
    import torch
    from crx import abi, synth, torch_api
    A, B = AB
    dev = torch.device("cuda", 0)
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
    # tracking NLP
    p = synth.cfg2_mpccbf(96, N=12, seed=21, safe_start=True)
    d = abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"])
    a = [t(p[k]) for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [t(p["n_obs"], torch.int32)]
    full = torch_api.cbf_solve_dev(d, *a)
    act = torch.from_numpy((np.arange(96) % 3 != 1).astype(np.int32)).to(dev)
    ws = torch_api.CbfWorkspace(d, 96, dev)
    ws.X.fill_(-7.0); ws.U.fill_(-7.0); ws.cost.fill_(-7.0)
    torch_api.cbf_solve_dev(d, *a, ws=ws, active=act)
    torch.cuda.synchronize()
    on = act.cpu().numpy() != 0
    assert (ws.status.cpu().numpy()[~on] == abi.CRX_SKIPPED).all() and (ws.iters.cpu().numpy()[~on] == 0).all()
    assert (ws.X.cpu().numpy()[~on] == -7.0).all() and (ws.U.cpu().numpy()[~on] == -7.0).all() and (ws.cost.cpu().numpy()[~on] == -7.0).all()
    for k in ("X", "U", "cost", "status", "iters", "kkt"):
        np.testing.assert_array_equal(getattr(ws, k).cpu().numpy()[on], getattr(full, k).cpu().numpy()[on])
    # learning-MPC QP
    dl, al = helpers.lmpc_inputs(golden_racing_game)
    n = al[0].shape[0]
    N = dl.N
    tl = [t(al[0]), t(al[1]), t(al[2]).reshape(n, N, 36), t(al[3]).reshape(n, N, 12), t(al[4]).reshape(n, N, 6), t(al[5]), t(al[6]),
          torch.full((n,), al[5].shape[2], dtype=torch.int32, device=dev)]
    fl = torch_api.lmpc_solve_dev(dl, *tl)
    act = torch.from_numpy((np.arange(n) % 2 == 0).astype(np.int32)).to(dev)
    wl = torch_api.LmpcWorkspace(dl, n, dev)
    wl.X.fill_(-7.0); wl.U.fill_(-7.0)
    torch_api.lmpc_solve_dev(dl, *tl, ws=wl, active=act)
    torch.cuda.synchronize()
    on = act.cpu().numpy() != 0
    assert (wl.status.cpu().numpy()[~on] == abi.CRX_SKIPPED).all() and (wl.X.cpu().numpy()[~on] == -7.0).all()
    for k in ("X", "U", "cost", "status", "iters"):
        np.testing.assert_array_equal(getattr(wl, k).cpu().numpy()[on], getattr(fl, k).cpu().numpy()[on])


def test_dispatch_order_is_bit_identical(gpu, golden_racing_game, AB):
    """crx_*_solve_ordered_dev: workgroup i solves problem order[i]; every problem gets exactly the bits the index-order launch
    gives it, with the mask and without (longest-first dispatch changes WHEN a problem runs, nothing else)."""
    import torch
    from crx import abi, synth, torch_api
    A, B = AB
    dev = torch.device("cuda", 0)
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
    p = synth.cfg4_tracking_cbf(3000, N=20, seed=31, safe_start=False)
    d = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    a = [t(p[k]) for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [t(p["n_obs"], torch.int32)]
    full = torch_api.cbf_solve_dev(d, *a)
    order = torch_api.longest_first(full.iters)
    o = order.cpu().numpy()
    np.testing.assert_array_equal(o, np.argsort(-np.minimum(full.iters.cpu().numpy(), 255), kind="stable"))   # a stable counting sort
    ws = torch_api.cbf_solve_dev(d, *a, order=order)
    rnd = torch.from_numpy(np.random.default_rng(3).permutation(3000).astype(np.int32)).to(dev)
    ws2 = torch_api.cbf_solve_dev(d, *a, order=rnd)
    for k in ("X", "U", "sigma", "cost", "status", "iters", "kkt"):
        np.testing.assert_array_equal(getattr(ws, k).cpu().numpy(), getattr(full, k).cpu().numpy())
        np.testing.assert_array_equal(getattr(ws2, k).cpu().numpy(), getattr(full, k).cpu().numpy())
    act = torch.from_numpy((np.arange(3000) % 3 != 1).astype(np.int32)).to(dev)
    wm = torch_api.CbfWorkspace(d, 3000, dev)
    wm.X.fill_(-7.0)
    om = torch_api.longest_first(full.iters, act)
    omc, itc = om.cpu().numpy(), full.iters.cpu().numpy()
    n_on = int((act != 0).sum().item())
    assert sorted(omc.tolist()) == list(range(3000)) and (act.cpu().numpy()[omc[:n_on]] != 0).all() and (np.diff(itc[omc[:n_on]]) <= 0).all()
    torch_api.cbf_solve_dev(d, *a, ws=wm, active=act, order=om)
    # the order with no previous solve (crx_cbf_order_dev): a permutation; cars that start inside a safety ellipse first (deepest
    # first), then those whose un-steered path enters one, then the rest -- checked against numpy on the same two barriers
    op = torch_api.cbf_order_dev(d, *a).cpu().numpy()
    live = np.arange(3)[None, :] < p["n_obs"][:, None]

    def barrier(j, s, ey):
        ds = (p["obs_s"][:, :, j] + p["lap_off"] - s[:, None]) / d.l_sum
        de = (p["obs_ey"][:, :, j] - ey[:, None]) / d.w_sum
        return np.where(live, ds ** d.degree + de ** d.degree - 1.0 - d.margin, 1e30).min(axis=1)
    h0 = barrier(0, p["x0"][:, 4], p["x0"][:, 5])
    hp = np.min([barrier(j, p["x0"][:, 4] + j * A[4, 0] * p["x0"][:, 0], p["xt"][:, j, 5]) for j in range(21)], axis=0)
    cls = np.where(h0 < 0, 0, np.where(hp < 0, 1, 2))
    assert sorted(op.tolist()) == list(range(3000)) and min((cls == c).sum() for c in range(3)) > 10
    assert (np.diff(cls[op]) < 0).sum() <= 3                 # a barrier within rounding of 0 may change class on the device
    n0 = int((cls == 0).sum())
    k0 = ((np.minimum(h0, 0.0) + 1.0 + d.margin) * (128.0 / (1.0 + d.margin))).astype(np.int64)
    assert (np.diff(k0[op[:n0 - 3]]) >= 0).sum() >= n0 - 10  # inside: 128 linear steps of h0, ascending
    wp = torch_api.cbf_solve_dev(d, *a, order=torch.from_numpy(op).to(dev))
    np.testing.assert_array_equal(wp.X.cpu().numpy(), full.X.cpu().numpy())
    # beyond the LDS key cache (65536 problems) the sort evaluates the key twice: same order as numpy on the same keys
    rep = 24
    big_it = full.iters.repeat(rep)
    ob = torch_api.longest_first(big_it).cpu().numpy()
    np.testing.assert_array_equal(ob, np.argsort(-np.minimum(big_it.cpu().numpy(), 255), kind="stable"))
    big = [x.repeat((rep,) + (1,) * (x.dim() - 1)).contiguous() for x in a]
    opb = torch_api.cbf_order_dev(d, *big).cpu().numpy()
    assert sorted(opb.tolist()) == list(range(3000 * rep))
    small = torch_api.cbf_order_dev(d, *[x[:60000].contiguous() for x in big]).cpu().numpy()      # cached path, same key code
    np.testing.assert_array_equal(small, opb[opb < 60000])
    # an obstacle-free batch through the C ABI with n_obs == NULL (include/crx.h admits it; ADVICE r4: the key kernel used to read it) and
    # device-resident dimensions that are garbage (zero / negative / NaN: the key falls back to the descriptor's pair like the solver does)
    import ctypes as C

    import crx
    from crx.torch_api import _ptr, _stream
    d0 = abi.cbf_desc(12, 0, A, B)
    o0 = torch.full((512,), -1, dtype=torch.int32, device=dev)
    x00, xt0 = a[0][:512].contiguous(), a[1][:512].contiguous()
    assert crx.lib().crx_cbf_order_dev(C.byref(d0), C.c_int(512), None, _ptr(x00), _ptr(xt0), None, None, None, None, None, _ptr(o0), _stream()) == 0, crx.lib().crx_last_error()
    torch.cuda.synchronize()
    assert sorted(o0.cpu().tolist()) == list(range(512))
    dims = torch.tensor([0.0, -1.0], dtype=torch.float64, device=dev).repeat(3000, d.n_obs_max, 1).contiguous()
    dims[::7, :, 0] = float("nan")
    np.testing.assert_array_equal(torch_api.cbf_order_dev(d, *a, obs_dims=dims).cpu().numpy(), op)
    # learning-MPC QP
    dl, al = helpers.lmpc_inputs(golden_racing_game)
    n = al[0].shape[0]
    N = dl.N
    tl = [t(al[0]), t(al[1]), t(al[2]).reshape(n, N, 36), t(al[3]).reshape(n, N, 12), t(al[4]).reshape(n, N, 6), t(al[5]), t(al[6]),
          torch.full((n,), al[5].shape[2], dtype=torch.int32, device=dev)]
    fl = torch_api.lmpc_solve_dev(dl, *tl)
    wl = torch_api.lmpc_solve_dev(dl, *tl, order=torch_api.longest_first(fl.iters))
    for k in ("X", "U", "lam", "cost", "status", "iters", "kkt"):
        np.testing.assert_array_equal(getattr(wl, k).cpu().numpy(), getattr(fl, k).cpu().numpy())


def test_zero_obstacle_nlps_incl_infeasible(gpu, orc, AB):
    """The 0-obstacle instantiation in controller mode (control.mpc_lti form, control.py:198-248, and mpc_multi_agents
    without vehicles in its window): 256 tracking problems on a narrow track (ey_max = 0.15) from states up to 0.14 off the
    centre line with headings up to 1.2 rad off -- a third cannot be brought back inside the input limits.  All rows are linear, so
    the infeasible ones end by box_certificate()'s proof; kernel and oracle must agree on every verdict and iteration
    count, and every verdict must agree with an LP solver on the feasible set."""
    from scipy.optimize import linprog
    from crx import abi
    A, B = AB
    N, Bn = 10, 256
    rng = np.random.default_rng(77)
    x0 = np.zeros((Bn, 6))
    x0[:, 0] = rng.uniform(0.8, 2.0, Bn); x0[:, 1] = rng.uniform(-0.2, 0.2, Bn); x0[:, 3] = rng.uniform(-1.2, 1.2, Bn)
    x0[:, 4] = rng.uniform(0, 5, Bn); x0[:, 5] = rng.uniform(-0.14, 0.14, Bn)
    xt = np.zeros((Bn, 6)); xt[:, 0] = 1.5
    d = abi.cbf_desc(N, 0, A, B, ey_max=0.15, v_min=0.5, v_max=2.2)
    z = np.zeros((Bn, 0, N + 1)); args = (x0, xt, z, z, np.zeros((Bn, 0)), np.zeros(Bn, dtype=np.int32))
    rg, ro = gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)
    _assert_same_verdicts("0-obstacle NLPs", rg, ro)
    inf = rg["status"] != 0
    assert 0.1 < inf.mean() < 0.7, inf.mean()
    assert rg["iters"][inf].mean() < 6.0                      # proven, not diverged
    ok = ~inf
    assert np.abs(rg["U"][ok] - ro["U"][ok]).max() <= 2e-5 and np.abs(rg["X"][ok] - ro["X"][ok]).max() <= 2e-5
    # LP check of every verdict: x_k = A^k x0 + sum_j A^(k-1-j) B u_j, v_min <= vx_k <= v_max, |ey_k| <= ey_max (k >= 1), u in its box
    Ap = [np.eye(6)]
    for _ in range(N):
        Ap.append(A @ Ap[-1])
    G = np.zeros((N + 1, 6, 2 * N))
    for k in range(1, N + 1):
        for j in range(k):
            G[k][:, 2 * j:2 * j + 2] = Ap[k - 1 - j] @ B
    bounds = [(-d.delta_max, d.delta_max), (-d.a_max, d.a_max)] * N
    wrong = []
    for b in range(Bn):
        rows, rhs = [], []
        for k in range(1, N + 1):
            f = Ap[k] @ x0[b]
            rows += [G[k][0], -G[k][0], G[k][5], -G[k][5]]
            rhs += [d.v_max - f[0], f[0] - d.v_min, d.ey_max - f[5], f[5] + d.ey_max]
        infeas0 = x0[b, 0] < d.v_min - 1e-8 or x0[b, 0] > d.v_max + 1e-8 or abs(x0[b, 5]) > d.ey_max + 1e-8     # quirk Q9
        lp = linprog(np.zeros(2 * N), A_ub=np.array(rows), b_ub=np.array(rhs), bounds=bounds, method="highs")
        if bool(inf[b]) != (lp.status == 2 or infeas0):
            lo = linprog(np.zeros(2 * N), A_ub=np.array(rows), b_ub=np.array(rhs) - 1e-7, bounds=bounds, method="highs").status == 2
            hi = linprog(np.zeros(2 * N), A_ub=np.array(rows), b_ub=np.array(rhs) + 1e-7, bounds=bounds, method="highs").status == 2
            if lo == hi:
                wrong.append((b, int(rg["status"][b]), int(rg["iters"][b]), int(lp.status)))
    assert not wrong, wrong


def test_packed_wave_reductions(gpu):
    """(and the DPP dot products of the sweeps, row_dot)  crx_wave.h wave_sum4 / wave_max4 / wave_sum2 / wave_max2 (v_permlane32_swap / v_permlane16_swap + one row reduction)
    on random 64-lane inputs through the hidden crx_debug_wave_reduce: the maxima exact, the sums to rounding of a
    different association, each value in ITS output slot (a transposed row map would swap slots 1 and 2)."""
    import ctypes as C

    import crx
    L = crx.lib()
    rng = np.random.default_rng(8)
    for trial in range(6):
        x = rng.normal(0, 10.0 ** rng.integers(-3, 4), (4, 64)) + np.array([[1.0], [-20.0], [300.0], [0.004]])
        if trial == 5:
            x = np.tile(np.arange(64.0), (4, 1)) * np.array([[1.0], [2.0], [3.0], [4.0]])     # exact sums 2016 k
        xin = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros(16 + 3 * 64)
        assert L.crx_debug_wave_reduce(xin.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
        # [r4] row_dot: broadcast-and-FMA in one v_fmac_f64_dpp row_newbcast -- lane i of the reader's OWN 16-lane row, under full EXEC, inside
        # a predicated region (lanes < 10, the sweeps' form), and chained on a value the previous dot product has just written
        a, b, c, d = x
        m = [(i + 1.0) * b + d for i in range(7)]
        row0 = (np.arange(64) // 16) * 16
        full = c + sum(m[i] * a[row0 + i] for i in range(7))
        scale = np.abs(c) + sum(np.abs(m[i] * a[row0 + i]) for i in range(7))
        assert (np.abs(out[16:80] - full) <= 16 * 2.3e-16 * scale).all(), np.abs(out[16:80] - full).max()
        msk = c[:10] + sum(m[i][:10] * a[3 + i] for i in range(7))
        assert (np.abs(out[80:90] - msk) <= 16 * 2.3e-16 * (np.abs(c[:10]) + sum(np.abs(m[i][:10] * a[3 + i]) for i in range(7)))).all() and np.isnan(out[90:144]).all()
        first = full[:10]
        chn = first + sum(m[i][:10] * first[7 + i] for i in range(3))
        assert (np.abs(out[144:154] - chn) <= 64 * 2.3e-16 * (np.abs(first) + sum(np.abs(m[i][:10] * first[7 + i]) for i in range(3)) + scale[:10])).all() and np.isnan(out[154:208]).all()
        ref_s, ref_m = x.sum(axis=1), x.max(axis=1)
        tol = 64 * 2.3e-16 * np.abs(x).sum(axis=1)
        assert (np.abs(out[0:4] - ref_s) <= tol).all(), (out[0:4], ref_s)
        np.testing.assert_array_equal(out[4:8], ref_m)
        assert (np.abs(out[8:10] - ref_s[:2]) <= tol[:2]).all()
        np.testing.assert_array_equal(out[10:12], ref_m[2:])
        assert abs(out[12] - ref_s[0]) <= tol[0] and out[13] == ref_m[1] and out[14] == x[2].min()
        if trial == 5:
            np.testing.assert_array_equal(out[0:4], 2016.0 * np.arange(1, 5))


def test_lmpc_addtraj_device(gpu, golden_racing_game):
    """crx_lmpc_addtraj_dev (add_trajectory on the device, masked by `crossed`) against the oracle, which
    tests/test_oracle_golden.py pins to the safe set the reference's own code built: everything it touches, bit for bit --
    races that crossed, races that did not, races whose safe set is full."""
    import ctypes as C

    import torch

    import oracle
    from crx import abi, torch_api
    from test_oracle_golden import _oracle_addtraj
    oracle.load()
    lib = C.CDLL(oracle._LIB)
    g = golden_racing_game
    Bn, P, L, Ls = 48, 600, 4, float(g["lap_length"])
    d = abi.lmpcprep_desc(12, P, L, 9, 0.1, Ls)
    rng = np.random.default_rng(12)
    ss, us, qf = rng.normal(size=(Bn, L, P, 6)), rng.normal(size=(Bn, L, P, 2)), rng.normal(size=(Bn, L, P))
    time_ss = rng.integers(100, 300, (Bn, L)).astype(np.int32)
    it = rng.integers(2, L + 1, Bn).astype(np.int32)                    # some races are full (it == L)
    step = rng.integers(0, 200, Bn).astype(np.int32)
    crossed = (rng.random(Bn) < 0.6).astype(np.int32)
    n_log = rng.integers(120, 320, Bn).astype(np.int32)
    log_x, log_u = rng.normal(size=(Bn, P, 6)), rng.normal(size=(Bn, P, 2))
    for b in range(Bn):                                                 # s rises along the lap and passes the line at the last sample
        log_x[b, :n_log[b], 4] = np.linspace(0.01, Ls + 0.05, n_log[b])
    x = rng.normal(size=(Bn, 6))
    host = [a.copy() for a in (log_x, log_u, n_log, ss, us, qf, time_ss, it, step)]
    st_o = _oracle_addtraj(lib, d, crossed, host[0], host[1], host[2], host[3], host[4], host[5], host[6], host[7], host[8], x)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    dv = [t(a) for a in (log_x, log_u, n_log, ss, us, qf, time_ss, it, step)]
    st_d = torch.zeros(Bn, dtype=torch.int32, device=dev)
    torch_api.lmpc_addtraj_dev(d, t(crossed), dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], dv[7], dv[8], t(x), st_d)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st_d.cpu().numpy(), st_o)
    for name, a, b in zip(("log_x", "log_u", "n_log", "ss", "us", "qf", "time_ss", "it", "step"), dv, host):
        np.testing.assert_array_equal(a.cpu().numpy(), b, err_msg=name)
    assert (st_o == 1).sum() >= 3 and ((crossed == 1) & (st_o == 0)).sum() >= 10 and (crossed == 0).sum() >= 10


def test_masked_planner_plan(gpu, AB):
    """crx_planner_plan_masked_dev: the region QPs of a masked-out scenario are skipped (status CRX_SKIPPED, X untouched),
    the other scenarios' QPs and winners are bit-identical to the unmasked launch."""
    import torch
    from crx import abi, synth, torch_api
    A, B = AB
    n_scen, N = 96, 12
    p = synth.cfg3_planner(n_scen, N=N, seed=31)
    V = p["V"]; R = V + 1
    d, sd = abi.planner_desc(N, A, B), abi.select_desc(N, V, p["lap_length"])
    dev = torch.device("cuda", 0)
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
    qin = [t(p[k]) for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")]
    sin = [t(p["n_veh"], torch.int32), t(p["obs_s"]), t(p["obs_ey"]), t(p["old_flag"], torch.int32)]

    def run(active):
        ws, sws = torch_api.PlannerWorkspace(d, n_scen * R, dev), torch_api.SelectWorkspace(sd, n_scen, dev)
        ws.X.fill_(-7.0); ws.U.fill_(-7.0)
        torch_api.planner_plan_dev(d, sd, *qin, *sin, ws, sws, active=active)
        torch.cuda.synchronize()
        return ws, sws

    full, fsel = run(None)
    act = torch.from_numpy((np.arange(n_scen) % 4 != 2).astype(np.int32)).to(dev)
    ws, sws = run(act)
    on = act.cpu().numpy() != 0
    onq = np.repeat(on, R)
    assert (ws.status.cpu().numpy()[~onq] == abi.CRX_SKIPPED).all() and (ws.X.cpu().numpy()[~onq] == -7.0).all()
    for k in ("X", "U", "cost", "status", "iters"):
        np.testing.assert_array_equal(getattr(ws, k).cpu().numpy()[onq], getattr(full, k).cpu().numpy()[onq], err_msg=k)
    np.testing.assert_array_equal(sws.flag.cpu().numpy()[on], fsel.flag.cpu().numpy()[on])
    np.testing.assert_array_equal(sws.best_X.cpu().numpy()[on], fsel.best_X.cpu().numpy()[on])


@pytest.mark.parametrize("eps", [1e-3, 1e-8, -1e-6])
def test_certificate_on_razor_thin_qps(gpu, orc, AB, eps):
    """The kernel's infeasibility proof on planner QPs feasible / infeasible by a hair (tests/helpers.py thin_corridor_qps):
    all converge for eps > 0, all are reported infeasible for eps < 0, verdict by verdict like the oracle."""
    d, args = helpers.thin_corridor_qps(orc, AB, eps)
    rg, ro = gpu.planner_solve(d, *args), orc.planner_solve(d, *args)
    if eps > 0:
        assert (rg["status"] == 0).all(), np.bincount(rg["status"], minlength=6)
    else:   # proved infeasible (2); a rare one ends by the divergence heuristic, which is not a proof (5: CRX_STALLED)
        assert np.isin(rg["status"], (2, 5)).all() and (rg["status"] == 2).mean() >= 0.99, np.bincount(rg["status"], minlength=6)
    np.testing.assert_array_equal(rg["status"], np.asarray(ro["status"]))
    assert np.abs(rg["iters"] - np.asarray(ro["iters"])).max() <= 2


# ---- the BASELINE draws as the reference itself builds them (tests/golden/cfg{2,3,4}_draw.npz, tests/test_draw_fixtures.py) ----
@pytest.mark.parametrize("kind,group", [("cfg2", "draw"), ("cfg2", "lapped"), ("cfg4", "draw"), ("cfg4", "lapped")])
@pytest.mark.parametrize("T", ["default", "tight"])
def test_reference_built_draws_cbf(gpu, orc, AB, kind, group, T):
    """libcrx on the problems bench.py times, against (i) the certified KKT points of the reference-built problems and (ii) the
    oracle, problem by problem: the three-way check of VERDICT r2 item 1 (the row-level identity synth+hostprep == reference
    is the CPU half, tests/test_draw_fixtures.py::test_cbf_rows_and_prep_match_reference)."""
    import test_draw_fixtures as tdf

    Tt = tdf.DEFAULT if T == "default" else tdf.TIGHT
    table, worst, rg, g, how = tdf._solve_and_compare(gpu, orc, AB, kind, group, Tt["tol"], Tt)
    n = len(how)
    assert table["converged_certified"] >= 0.8 * n, table
    # (a converged solve without a certified point on record: at 1e-8 a problem no candidate reaches at 1e-11; at 1e-11 one the
    # kernel's Riccati recursion still reaches where the oracle's condensed Cholesky stalls -- `tight_stall` below)
    assert table["converged_uncertified"] <= max(1, n // 50), table
    # and the oracle on the same inputs: same verdicts / iteration counts up to the classified exceptions
    A, B = AB
    p = tdf.cbf_batch(kind, group == "lapped")
    d = tdf.cbf_desc(kind, A, B, Tt["tol"])
    idx = g["index"].astype(int)
    ro = orc.cbf_solve(d, *[p[k][idx] for k in tdf.KEYS])
    d.opts.restore_iters = -1
    g0, o0 = gpu.cbf_solve(d, *[p[k][idx] for k in tdf.KEYS]), orc.cbf_solve(d, *[p[k][idx] for k in tdf.KEYS])
    # (N = 20 at tol 1e-11: the oracle's condensed Cholesky stalls at 2e-9..8e-9 on 2 of the 48 problems the Riccati recursion takes to 2e-12)
    stall = max(2, n // 16)
    _assert_same_verdicts("%s/%s no restoration" % (kind, group), g0, o0, tol=Tt["tol"], max_tight_stall=stall)
    touched = set(np.nonzero((g0["status"] != rg["status"]) | (g0["iters"] != rg["iters"]) | (o0["status"] != ro["status"]) | (o0["iters"] != ro["iters"]))[0].tolist())
    _assert_same_verdicts("%s/%s" % (kind, group), rg, ro, tol=Tt["tol"], restored=frozenset(touched), max_restored_verdict=max(2, len(touched) // 5),
                          max_tight_stall=stall)


@pytest.mark.parametrize("cars", ["three_cars", "five_cars"])
@pytest.mark.parametrize("T", ["default", "tight"])
def test_reference_built_draw_planner(gpu, orc, AB, T, cars):
    """cfg3: the region QPs of the first 64 scenarios of the BASELINE draw as the reference's get_local_traj builds them --
    verdict of every region = HiGHS on the reference's rows, trajectories = the certified minimisers, direction flag and
    winning trajectory = the reference's.  [r5] five_cars: 24 scenarios with FIVE vehicles of interest = six regions each
    (tests/golden/cfg3_many.npz; CRX_MAX_VEH = 6), host prep and device prep alike."""
    import test_draw_fixtures as tdf

    Tt = tdf.DEFAULT if T == "default" else tdf.TIGHT
    draw = tdf.PLANNER_DRAWS[0 if cars == "three_cars" else 1]
    V = draw[1]
    rg, g = tdf._planner_compare(gpu, orc, AB, Tt, draw)
    # selection on the device as well (a flag may differ from the reference's only inside a tie of two identical regions: _planner_compare)
    from crx import abi
    idx = g["index"].astype(int)
    p = tdf.planner_batch(V, draw[2])
    n = len(idx)
    sd = abi.select_desc(12, V, tdf.LAP)
    sel = gpu.select(sd, p["n_veh"][idx], rg["X"].reshape(n, V + 1, 13, 6), p["obs_s"][idx], p["obs_ey"][idx], p["old_flag"][idx])
    so = orc.select(sd, p["n_veh"][idx], rg["X"].reshape(n, V + 1, 13, 6), p["obs_s"][idx], p["obs_ey"][idx], p["old_flag"][idx])
    np.testing.assert_array_equal(sel["flag"], so["flag"])
    assert (sel["flag"] != g["direction_flag"]).sum() <= max(1, n // 20)
    if cars == "five_cars" and T == "default":
        # the device-side prep (Bezier polylines, ey bounds for six regions) = the host mirror's, which the reference pins (test_draw_fixtures)
        A, B = AB
        raw = p["raw"]
        pd = abi.prep_desc(12, V, len(raw["opt_s"]), float(raw["track_width"]), float(raw["lap_length"]))
        dp = gpu.planner_prep(pd, raw["x"][idx], raw["x"][idx], raw["n_veh"][idx], raw["veh_info"][idx], raw["max_dv"][idx], raw["obs_s"][idx],
                              raw["obs_ey"][idx], raw["opt_s"], raw["opt_ey"])
        rows = np.concatenate([np.arange(b * (V + 1), (b + 1) * (V + 1)) for b in idx])
        for k, tol in (("bez_s", 1e-12), ("bez_ey", 1e-12), ("ey_lb", 1e-13), ("ey_ub", 1e-13), ("x0", 0.0)):
            assert np.abs(np.asarray(dp[k]) - p[k][rows]).max() <= tol, k


def test_allgather_winners_c_abi_one_rank(gpu):
    """crx_allgather_winners_dev (include/crx.h): libcrx's own RCCL communicator at world size 1 -- id, init, pack + ncclAllGather on
    a torch stream, destroy.  More ranks need more GPUs than a test box has; the record layout and the padding are checked."""
    import ctypes as C

    import torch

    import crx
    from crx import dist as cdist
    from crx.torch_api import _ptr, _stream

    L = crx.lib()
    cdist.CrxComm.ensure()
    assert L.crx_comm_world() == 1 and L.crx_comm_rank() == 0
    n, n_max, N = 37, 40, 12
    rec = 1 + 6 * (N + 1)
    dev = torch.device("cuda", 0)
    flag = torch.arange(n, dtype=torch.int32, device=dev) % 4
    X = torch.randn((n, N + 1, 6), dtype=torch.float64, device=dev)
    send = torch.full((n_max, rec), 7.0, dtype=torch.float64, device=dev)
    recv = torch.full((n_max, rec), -1.0, dtype=torch.float64, device=dev)
    status = (torch.arange(n, dtype=torch.int32, device=dev) * 7) % 6                 # every crx_status value
    assert L.crx_allgather_winners_dev(C.c_int(n), C.c_int(n_max), C.c_int(N), _ptr(flag), _ptr(status), _ptr(X), _ptr(send), _ptr(recv), _stream()) == 0, L.crx_last_error()
    torch.cuda.synchronize()
    head = recv[:, :1].contiguous().view(torch.int32)                                  # {int32 flag; int32 status; double X[N+1][6]}
    assert torch.equal(head[:n, 0], flag) and torch.equal(head[:n, 1], status) and torch.equal(recv[:n, 1:], X.reshape(n, -1))
    assert (recv[n:] == 0).all()
    assert L.crx_allgather_winners_dev(C.c_int(n), C.c_int(n_max), C.c_int(N), _ptr(flag), None, _ptr(X), _ptr(send), _ptr(recv), _stream()) == 0   # NULL status: field 0
    torch.cuda.synchronize()
    head = recv[:, :1].contiguous().view(torch.int32)
    assert torch.equal(head[:n, 0], flag) and not head[:, 1].any()
    # the product path that uses it (crx.dist.WinnerExchange with COLLECTIVE = "crx")
    cdist.COLLECTIVE, cdist.FORCE_COLLECTIVE = "crx", True
    try:
        ex = cdist.WinnerExchange(n, N, dev)
        f2, X2, s2 = ex(flag, X, status)
        torch.cuda.synchronize()
        assert torch.equal(f2, flag) and torch.equal(X2, X) and torch.equal(s2, status)
    finally:
        cdist.COLLECTIVE, cdist.FORCE_COLLECTIVE = "torch", False
        cdist.CrxComm.destroy()
    assert L.crx_comm_world() == 0


@pytest.mark.parametrize("T", ["default", "tight"])
def test_unequal_obstacle_sizes(gpu, orc, AB, T):
    """crx_cbf_solve_dims: obstacle vehicles of different sizes (control.py:529-535 reads every obstacle's own CarParam) -- libcrx
    against the certified KKT points of the problems the reference built for two cars of random dimensions (tests/golden/cfg2_dims.npz)
    and against the oracle; the device entry point with per-problem dims equals the host one."""
    import torch

    import test_draw_fixtures as tdf
    from crx import abi, torch_api

    Tt = tdf.DEFAULT if T == "default" else tdf.TIGHT
    rg = tdf._dims_compare(gpu, orc, AB, Tt)
    A, B = AB
    g, p = tdf.dims_batch()
    d = abi.cbf_desc(12, 2, A, B, alpha=0.8, margin=0.2)
    d.opts.tol = Tt["tol"]
    ro = orc.cbf_solve(d, *[p[k] for k in tdf.KEYS], obs_dims=p["obs_dims"])
    d.opts.restore_iters = -1
    g0 = gpu.cbf_solve(d, *[p[k] for k in tdf.KEYS], obs_dims=p["obs_dims"])
    o0 = orc.cbf_solve(d, *[p[k] for k in tdf.KEYS], obs_dims=p["obs_dims"])
    _assert_same_verdicts("unequal cars, no restoration", g0, o0, tol=Tt["tol"], max_tight_stall=2)
    touched = frozenset(np.nonzero((g0["status"] != rg["status"]) | (g0["iters"] != rg["iters"]) | (o0["status"] != ro["status"]) | (o0["iters"] != ro["iters"]))[0].tolist())
    _assert_same_verdicts("unequal cars", rg, ro, tol=Tt["tol"], restored=touched, max_restored_verdict=2, max_tight_stall=2)
    # device entry point
    d.opts.restore_iters = abi.default_opts().restore_iters
    dev = torch.device("cuda", 0)
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)   # noqa: E731
    ws = torch_api.cbf_solve_dev(d, t(p["x0"]), t(p["xt"]), t(p["obs_s"]), t(p["obs_ey"]), t(p["lap_off"]), t(p["n_obs"], torch.int32), obs_dims=t(p["obs_dims"]))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(ws.X.cpu().numpy(), rg["X"])
    np.testing.assert_array_equal(ws.status.cpu().numpy(), rg["status"])


@pytest.mark.parametrize("T", ["default", "tight"])
def test_many_obstacles(gpu, orc, AB, T):
    """[r4] CRX_MAX_OBS = 6 (control.py:524-562 admits any number of vehicles in the window): four and five obstacle cars on the generic
    six-obstacle instantiation of the solver kernel -- against the certified KKT points of the problems the reference itself built
    (tests/golden/cfg2_many.npz) and against the oracle, verdict by verdict; and a three-car problem padded to six slots gives the
    answer of the tuned three-obstacle instantiation (same KKT point through another instantiation)."""
    import test_draw_fixtures as tdf
    from crx import abi, synth

    Tt = tdf.DEFAULT if T == "default" else tdf.TIGHT
    rg = tdf._many_compare(gpu, AB, Tt)
    A, B = AB
    g, p = tdf.many_batch()
    d = abi.cbf_desc(12, 6, A, B, alpha=0.8, margin=0.2)
    d.opts.tol = Tt["tol"]
    ro = orc.cbf_solve(d, *[p[k] for k in tdf.KEYS])
    _assert_same_verdicts("many cars", rg, ro, tol=Tt["tol"], max_tight_stall=2, max_other=1)
    _cmp("many cars", rg, ro, need_same_status=False, T=Tt if T == "default" else dict(Tt, x=Tt["loose"]["x"], u=Tt["loose"]["u"], xw=Tt["loose"]["xw"]))
    # three obstacles through six slots == three obstacles through three slots
    q = synth.cfg4_tracking_cbf(32, N=12, seed=9)
    kw = dict(alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    d3, d6 = abi.cbf_desc(12, 3, A, B, **kw), abi.cbf_desc(12, 6, A, B, **kw)
    pad = lambda a: np.concatenate([a, np.zeros((a.shape[0], 3) + a.shape[2:])], axis=1)   # noqa: E731
    r3 = gpu.cbf_solve(d3, q["x0"], q["xt"], q["obs_s"], q["obs_ey"], q["lap_off"], q["n_obs"])
    r6 = gpu.cbf_solve(d6, q["x0"], q["xt"], pad(q["obs_s"]), pad(q["obs_ey"]), pad(q["lap_off"]), q["n_obs"])
    both = (r3["status"] == 0) & (r6["status"] == 0)
    assert both.sum() >= 28 and (r3["status"] == 0).sum() == (r6["status"] == 0).sum()
    assert np.abs(r3["X"][both] - r6["X"][both]).max() <= 1e-4 and np.abs(r3["cost"][both] - r6["cost"][both]).max() <= 1e-6 * np.abs(r3["cost"][both]).max()
    # the largest instantiation, <6, 24> (109 KB of LDS, one problem per CU): horizon 24, three cars in six slots, against the oracle
    q = synth.cfg4_tracking_cbf(16, N=24, seed=11)
    d6 = abi.cbf_desc(24, 6, A, B, **kw)
    d6.opts.tol = Tt["tol"]
    a6 = (q["x0"], q["xt"], pad(q["obs_s"]), pad(q["obs_ey"]), pad(q["lap_off"]), q["n_obs"])
    rg, ro = gpu.cbf_solve(d6, *a6), orc.cbf_solve(d6, *a6)
    _assert_same_verdicts("six slots, N = 24", rg, ro, tol=Tt["tol"], max_tight_stall=1, max_other=1)
    both = (rg["status"] == 0) & (ro["status"] == 0)
    assert both.sum() >= 12 and np.abs(rg["cost"][both] - ro["cost"][both]).max() <= 1e-6 * np.abs(ro["cost"][both]).max()
