"""Shared builders: turn a golden scenario into crx C-ABI inputs through the PRODUCT's host prep."""
import numpy as np
from crx import abi, hostprep


def mpccbf_inputs(g, A, B, n_obs_max=None):
    N = int(g["N"])
    V = g["obs_pred"].shape[0]
    obs_s = g["obs_pred"][None, :, 4, :]
    obs_ey = g["obs_pred"][None, :, 5, :]
    x0 = g["x0"][None, :]
    keep, lap_off = hostprep.cbf_window(x0, obs_s[:, :, 0], float(g["lap_length"]))
    nmax = max(1, int(keep.sum())) if n_obs_max is None else n_obs_max
    ps, pe, po, n = hostprep.pack_obstacles(keep, obs_s, obs_ey, lap_off, nmax)
    d = abi.cbf_desc(N, nmax, A, B, alpha=float(g["alpha"]), margin=0.2, ey_max=float(g["width"]))
    xt = np.array([[float(g["vt"]), 0, 0, 0, 0, 0.0]])
    return d, (x0, xt, ps, pe, po, n)


def planner_inputs(g, A, B):
    N = int(g["N"])
    V = g["obs_pred"].shape[0]
    R = V + 1
    obs_s = g["obs_pred"][None, :, 4, :]
    obs_ey = g["obs_pred"][None, :, 5, :]
    lb, ub = hostprep.planner_ey_bounds(g["x_wrapped"][None, :], obs_s, obs_ey, np.array([V]),
                                        float(g["width"]), float(g["lap_length"]), N)
    d = abi.planner_desc(N, A, B)
    x0 = np.repeat(g["x_raw"][None, :], R, axis=0)
    return d, (x0, g["bezier_xcurvs"][:, :, 0], g["bezier_xcurvs"][:, :, 1], lb[0], ub[0])


def mma_inputs(g, A, B):
    """mpc_multi_agents (control.py:251-473): Q/alpha/margin literals, per-stage target."""
    N = int(g["N"])
    obs_s = g["mma_obs_pred"][None, :, 4, :]
    obs_ey = g["mma_obs_pred"][None, :, 5, :]
    x = g["x_wrapped"][None, :]
    keep, lap_off = hostprep.cbf_window(x, obs_s[:, :, 0], float(g["lap_length"]))
    nmax = max(1, int(keep.sum()))
    ps, pe, po, n = hostprep.pack_obstacles(keep, obs_s, obs_ey, lap_off, nmax)
    d = abi.cbf_desc(N, nmax, A, B, Q=(10.0, 0, 0, 5.0, 0, 50.0), alpha=0.6, margin=0.15,
                     ey_max=float(g["width"]), per_stage_target=True)
    xt = hostprep.tracking_targets(g["x_wrapped"], g["traj_xcurv"], N)[None]
    return d, (x, xt, ps, pe, po, n)


def lmpc_inputs(g, idx=None):
    """Learning-MPC QPs recorded from the reference's control.lmpc (tests/golden/racing_game.npz)."""
    sl = slice(None) if idx is None else idx
    d = abi.lmpc_desc(N=g["lmpc/A"].shape[1], n_ss_max=g["lmpc/ss"].shape[2], ey_max=float(g["lmpc/lap_width"][0]))
    return d, (g["lmpc/x"][sl], g["lmpc/u_old"][sl], g["lmpc/A"][sl], g["lmpc/B"][sl], g["lmpc/C"][sl],
               g["lmpc/ss"][sl], g["lmpc/qfun"][sl])


def game_draw_inputs():
    """tests/golden/game_draw.npz (tests/golden/tools/make_draws.py game): the learning-MPC QPs the REFERENCE builds -- its own regression,
    safe-set selection and control.lmpc under the recording stand-in -- at the states the benched closed loop (bench.py `game`) visits in
    front of control steps 0 / 20 / 40 / 60 / 80 of 32 races, with HiGHS's verdict on each recorded QP."""
    import os

    import conftest
    g = np.load(os.path.join(conftest.ROOT, "tests", "golden", "game_draw.npz"))
    d = abi.lmpc_desc(N=g["A"].shape[1], n_ss_max=g["ss"].shape[2], ey_max=1.0)
    n = g["x"].shape[0]
    return g, d, (g["x"], g["u_old"], g["A"].reshape(n, -1, 36), g["B"].reshape(n, -1, 12), g["C"], g["ss"], g["qfun"])


def check_game_draw(r, g):
    """One solver's answers on the game draw: converged exactly where HiGHS finds the reference's QP feasible, PROVED infeasible (status 2:
    the relaxed-x0 plan is what comes back) where HiGHS says infeasible; the unique solution where the third solver certified one."""
    feas = g["lp_status"] != 2
    assert (g["lp_status"][feas] == 0).all()
    st = np.asarray(r["status"])
    assert ((st == 0) == feas).all(), np.nonzero((st == 0) != feas)[0]
    assert (st[~feas] == 2).all(), np.bincount(st[~feas], minlength=6)          # every one by a proof (screen / Farkas certificate), none by a heuristic
    ok = g["success"] & feas
    assert ok.sum() >= 70
    # (default tolerance 1e-8 against points certified at 1e-11; mid-lap stage models are worse conditioned than the recorded lap's: 5e-6 there)
    assert np.abs(np.asarray(r["X"])[ok] - g["X"][ok]).max() <= 5e-5 and np.abs(np.asarray(r["U"])[ok] - g["U"][ok]).max() <= 5e-4
    # the fixture also records what the DEVICE loop had made of the same states, from raw state through its own regression: the same verdicts,
    # the same 44 safe-set points, stage models within 1e-3 (cond 3e11 normal matrices)
    assert ((g["dev_status"] != 0) == ~feas).all() and g["dev_ss_equal"].all() and g["dev_q_equal"].all() and g["dev_model_dev"].max() <= 2e-3
    # [r6] the RELAXED plan of the infeasible ones (libcrx's own semantics: x_0 = xcurv + w, cost += 1e4 w'w, no rows on stage 0 -- a strictly convex QP)
    # against the third solver's certified solution of that QP, written down explicitly from the reference's problem data
    # (tests/golden/tools/lmpc_relaxed.py; the same explicit form reproduces the reference-built feasible QPs to 6e-7)
    rel = g["relaxed_ok"]
    assert rel.sum() == (~feas).sum() >= 60 and (rel == ~feas).all()
    X, U = np.asarray(r["X"])[rel], np.asarray(r["U"])[rel]
    assert np.abs(X - g["relaxed_X"][rel]).max() <= 5e-5 and np.abs(U - g["relaxed_U"][rel]).max() <= 5e-5
    assert np.abs((X[:, 0] - g["x"][rel]) - g["relaxed_w"][rel]).max() <= 1e-6          # the plan starts at xcurv + w
    return int(feas.sum()), int((~feas).sum())


def path_inputs(c):
    """crx_path_solve inputs of a recorded OvertakePathPlanner scenario (tests/golden/path_planner.npz), through the
    PRODUCT's host prep (planning/overtake_path_planner.path_qp_inputs)."""
    import os

    import conftest
    from planning import overtake_path_planner as opp
    from planning import planner_helper as ph

    opt = np.genfromtxt(os.path.join(conftest.ROOT, "data/optimal_traj/xcurv_l_shape.csv"), delimiter=",")
    N = int(c["N"])
    names = [str(x) for x in c["veh_names"]]
    interest = [n for n, f in zip(names, c["veh_is_interest"]) if f]
    xc = {n: c["veh_xcurv"][i] for i, n in enumerate(names)}
    cars = {n: c["cars"][i] for i, n in enumerate(names)}
    order = ph.sort_by_ey(interest, lambda n: xc[n][5])
    assert order == [str(x) for x in c["sorted_vehicles"]]
    x = c["x"]
    info = ph.AgentInfo()
    vx = np.array([xc[n][0] for n in order]); s = np.array([xc[n][4] for n in order])
    dv = np.abs(x[0] - vx)
    L = float(c["lap_length"])
    s = np.where(s <= 20, s + L, s)
    info.min_vx, info.max_vx, info.min_delta_v, info.max_delta_v = vx.min(), vx.max(), dv.min(), dv.max()
    info.min_s, info.max_s = hostprep.wrap_above(s.min(), L), hostprep.wrap_above(s.max(), L)
    obs_infos = np.array([[xc[n][4], cars[n][2], cars[n][2]] for n in order])       # scripted cars: ey constant
    cps = ph.bezier_control_points(len(order), obs_infos, info.max_delta_v, 0.5, float(c["width"]), L, 0.2, opt, x)
    bez = ph.bezier_polylines(cps, N)
    qp = opp.path_qp_inputs(x[4], x[5], obs_infos, info, cps, bez, opt[:, 4], opt[:, 5], N, float(c["width"]), L, 4.5, 0.5, 0.4, 0.2)
    return abi.path_desc(N, float(c["alpha"])), qp


def lmpc_lap_setup(g, track):
    """The state of the reference's racing game at the start of its first learning-MPC lap, from tests/golden/racing_game.npz,
    in libcrx's lap-major safe-set layout: (prep desc, ss [L,P,6], us [L,P,2], qf [L,P], time_ss [L], lin_points [N+1,6],
    lin_input [N,2])."""
    N = 12
    ss = np.ascontiguousarray(g["ss/ss0"].transpose(2, 0, 1))
    us = np.ascontiguousarray(g["ss/u0"].transpose(2, 0, 1))
    qf = np.ascontiguousarray(g["ss/Qfun0"].T)
    time_ss = g["ss/time_ss"].astype(np.int32)
    tab = track.point_and_tangent
    d = abi.lmpcprep_desc(N, ss.shape[1], ss.shape[0], tab.shape[0], float(g["timestep"]), float(g["lap_length"]))
    # LMPCRacingGame.add_trajectory hands over the linearisation points of lap 0 (utils/base.py:651-653)
    return d, ss, us, qf, time_ss, ss[0, 1:N + 2].copy(), us[0, 1:N + 1].copy()


def random_scenes(S, VA, N, lap_length, seed=0):
    """Random planner scenes for the scene stage: ego + up to VA other vehicles around it (ahead, behind, across the start
    line), constant-speed predictions.  Returns (ego_xcurv, n_all, veh_xcurv, pred_s, pred_ey)."""
    rng = np.random.default_rng(seed)
    ego = np.zeros((S, 6))
    ego[:, 0] = rng.uniform(0.5, 1.6, S)
    ego[:, 4] = rng.uniform(0.0, lap_length * 1.02, S)        # some egos just past the lap length (raw state before the wrap)
    ego[:, 5] = rng.uniform(-0.6, 0.6, S)
    n_all = rng.integers(0, VA + 1, S).astype(np.int32)
    veh = np.zeros((S, VA, 6))
    veh[:, :, 0] = rng.uniform(0.3, 1.4, (S, VA))
    veh[:, :, 4] = (ego[:, 4, None] + rng.uniform(-1.0, 4.0, (S, VA))) % lap_length
    veh[:, :, 5] = 0.7 - 0.1 * rng.integers(0, 15, (S, VA))   # lanes repeat: ties in the partial sort
    j = np.arange(N + 1)
    pred_s = veh[:, :, 4, None] + 0.1 * j[None, None, :] * veh[:, :, 0, None]
    pred_ey = veh[:, :, 5, None] + 0.01 * np.sin(j[None, None, :] + veh[:, :, 4, None])
    return ego, n_all, veh, pred_s, pred_ey


def thin_corridor_qps(orc, AB, eps, n_scen=128, seed=5):
    """Planner QPs whose ey corridor is a tube of half-width eps around a trajectory known to be feasible (the solution of the
    original QP): feasible with margin eps for eps > 0, infeasible by |eps| for eps < 0.  Returns (desc, inputs)."""
    from crx import abi, synth
    A, B = AB
    N = 12
    p = synth.cfg3_planner(n_scen, N=N, seed=seed)
    d = abi.planner_desc(N, A, B)
    keys = ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")
    r0 = orc.planner_solve(d, *[p[k] for k in keys])
    ok = np.asarray(r0["status"]) == 0
    ey = np.asarray(r0["X"])[ok][:, :, 5]
    q = {k: p[k][ok].copy() for k in keys}
    q["ey_ub"] = ey[:, 0:N].max(axis=1) + eps                 # one upper bound per problem (overtake_traj_planner.py:277-324)
    q["ey_lb"][:, 1:] = ey[:, 1:N] - eps
    q["ey_lb"][:, 0] = np.minimum(q["ey_lb"][:, 0], ey[:, 0] - abs(eps))   # the row on the fixed x0 stays satisfied (quirk Q9)
    return d, tuple(q[k] for k in keys)


# ---- row probes of the oracle (oracle/crx_oracle.c crx_oracle_cbf_probe / crx_oracle_planner_probe) ------------------------
def _dp(a):
    import ctypes as C
    return a.ctypes.data_as(C.c_void_p)


def oracle_cbf_probe(orc, d, x0, xt, obs_s, obs_ey, lap_off, n_obs, U, sigma, obs_dims=None):
    """The cost, the UNSCALED CBF row values [n_obs_max][N], the variable boxes and the rolled-out states of the problem the
    oracle builds from one problem's C-ABI inputs, at the point (U [N][2], sigma [n_obs][N+1])."""
    import ctypes as C
    N, V = d.N, d.n_obs_max
    f64 = np.float64
    x0, xt, U = (np.ascontiguousarray(a, dtype=f64) for a in (x0, xt, U))
    obs_s, obs_ey, lap_off = (np.ascontiguousarray(a, dtype=f64).reshape(max(V, 1), -1) for a in (obs_s, obs_ey, lap_off))
    sig = np.zeros((max(V, 1), N + 1))
    sig[:n_obs] = np.asarray(sigma, dtype=f64).reshape(n_obs, N + 1)
    cost = C.c_double(0.0)
    rows, box, X = np.zeros((max(V, 1), N)), np.zeros(4 + 4 * (N + 1)), np.zeros((N + 1, 6))
    fn = orc.lib.crx_oracle_cbf_probe
    fn.restype = C.c_int
    dims = None if obs_dims is None else np.ascontiguousarray(obs_dims, dtype=f64).reshape(max(V, 1), 2)
    rc = fn(C.byref(d), _dp(x0), _dp(xt), _dp(obs_s), _dp(obs_ey), _dp(lap_off), C.c_int(int(n_obs)), _dp(dims) if dims is not None else None, _dp(U), _dp(sig), C.byref(cost),
            _dp(rows), _dp(box), _dp(X))
    assert rc == 0, rc
    b = box[4:].reshape(4, N + 1)
    return dict(cost=cost.value, cbf=rows[:n_obs], ulo=box[0:2], uhi=box[2:4], vlo=b[0], vhi=b[1], elo=b[2], ehi=b[3], X=X)


def oracle_planner_probe(orc, d, x0, bez_s, bez_ey, ey_lb, ey_ub, U):
    import ctypes as C
    N = d.N
    f64 = np.float64
    x0, bez_s, bez_ey, ey_lb, U = (np.ascontiguousarray(a, dtype=f64) for a in (x0, bez_s, bez_ey, ey_lb, U))
    cost = C.c_double(0.0)
    box, X = np.zeros(4 + 4 * (N + 1)), np.zeros((N + 1, 6))
    fn = orc.lib.crx_oracle_planner_probe
    fn.restype = C.c_int
    rc = fn(C.byref(d), _dp(x0), _dp(bez_s), _dp(bez_ey), _dp(ey_lb), C.c_double(float(ey_ub)), _dp(U), C.byref(cost), _dp(box), _dp(X))
    assert rc == 0, rc
    b = box[4:].reshape(4, N + 1)
    return dict(cost=cost.value, ulo=box[0:2], uhi=box[2:4], vlo=b[0], vhi=b[1], elo=b[2], ehi=b[3], X=X)
