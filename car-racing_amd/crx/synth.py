"""Synthetic batched racing scenarios for the BASELINE.json configs.

Deterministic: numpy Generator(PCG64(seed)), seed = config index.  Used by bench.py and by the
parity tests; no reference code or data beyond the LTI model CSVs is involved.
"""
import os

import numpy as np

LAP_L_SHAPE = 19.22957795362994  # l_shape lap length (SURVEY.md section 8c probe)
_ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def load_AB():
    A = np.genfromtxt(os.path.join(_ROOT, "data/sys/LTI/matrix_A.csv"), delimiter=",")
    B = np.genfromtxt(os.path.join(_ROOT, "data/sys/LTI/matrix_B.csv"), delimiter=",")
    return A, B


def _ego(rng, n, vx_lo, vx_hi):
    x = np.zeros((n, 6))
    x[:, 0] = rng.uniform(vx_lo, vx_hi, n)
    x[:, 1] = rng.normal(0.0, 0.02, n)
    x[:, 2] = rng.normal(0.0, 0.02, n)
    x[:, 3] = rng.uniform(-0.1, 0.1, n)
    x[:, 4] = rng.uniform(0.0, 19.2, n)
    x[:, 5] = rng.uniform(-0.6, 0.6, n)
    return x


def _safe_start(x0, obs_s, obs_ey, margin, l_sum=0.4, w_sum=0.2, degree=6, min_ttc=1.0):
    """Scenario filter: (i) h_0 >= 0 for every obstacle -- the CBF formulation presumes the current
    state is in the safe set (two 0.4 x 0.2 m cars cannot overlap; control.py:544-550); (ii) the ego is
    not already doomed: an obstacle in its lane and ahead must be at least `min_ttc` seconds of closing
    speed away from the unsafe set.  A 10 Hz controller that had been running would never be handed
    such a state; without (ii) ~1 % of the draws are unavoidable crashes whose NLP needs slacks of
    1e3..1e5 and 60+ interior-point iterations, and they alone set the latency of a 256-problem batch."""
    ds = x0[:, 4, None] - obs_s[:, :, 0]
    de = x0[:, 5, None] - obs_ey[:, :, 0]
    h0 = (ds / l_sum) ** degree + (de / w_sum) ** degree - 1.0 - margin
    ok = (h0 >= 0.05).all(axis=1)
    v_o = (obs_s[:, :, 1] - obs_s[:, :, 0]) / 0.1
    closing = x0[:, 0, None] - v_o
    gap = -ds - l_sum * (1.0 + margin) ** (1.0 / degree)          # ahead of the ego, to the unsafe set
    in_lane = np.abs(de) < w_sum * 1.25
    doomed = in_lane & (ds < 0) & (closing > 0) & (gap < min_ttc * closing)
    return ok & ~doomed.any(axis=1)


def _resample_unsafe(gen, batch, margin, max_rounds=64):
    """Draw with `gen(n)` until `batch` scenarios with a safe start are collected (rejection)."""
    parts = None
    have = 0
    for _ in range(max_rounds):
        p = gen(2 * batch)
        ok = _safe_start(p["x0"], p["obs_s"], p["obs_ey"], margin)
        p = {k: v[ok] for k, v in p.items()}
        parts = p if parts is None else {k: np.concatenate([parts[k], p[k]]) for k in p}
        have = parts["x0"].shape[0]
        if have >= batch:
            break
    return {k: v[:batch] for k, v in parts.items()}


def _lanes(rng, shape):
    return 0.7 - 0.1 * rng.integers(0, 15, size=shape)  # mirrors overtake_planner_test.py:81-82


def _lap_some(x0, lapped_frac, seed):
    """Put a fraction of the egos one lap ahead of the traffic (raw s in [lap, 2 lap): the state control.mpccbf sees on
    the step the ego crosses the line before utils/base.py:809 resets it, or when the caller counts s cumulatively).
    The window test works on wrapped distances (control.py:499-523) so the same obstacles stay in; their CBF rows get
    lap_off = lap_length for `diffs` and NONE for `diffs_next` (quirk Q1, control.py:539-542)."""
    if lapped_frac <= 0.0:
        return x0
    rng = np.random.default_rng(np.random.PCG64(1000 + seed))
    x0 = x0.copy()
    x0[rng.uniform(size=x0.shape[0]) < lapped_frac, 4] += LAP_L_SHAPE
    return x0


def _through_window(p, n_obs, lap_length):
    """What control.mpccbf / mpc_multi_agents do with the cars before building rows (control.py:499-523 / :293-309): the
    +-2 vx window on lap-wrapped distances decides which cars become obstacles, in dict order, and fixes their lap offsets.
    The product's host prep (hostprep.cbf_window / pack_obstacles) is used, as by control.mpccbf's mirror; the parity of
    that prep with the reference's own rows on these very draws is pinned by tests/golden/cfg{2,4}_draw.npz."""
    from . import hostprep

    keep, lap_off = hostprep.cbf_window(p["x0"], p["obs_s"][:, :, 0], lap_length)
    ps, pe, po, n = hostprep.pack_obstacles(keep, p["obs_s"], p["obs_ey"], lap_off, n_obs)
    p.update(obs_s=ps, obs_ey=pe, lap_off=po, n_obs=n)
    return p


def cfg2_mpccbf(batch=256, N=12, seed=2, n_obs=1, safe_start=True, lapped_frac=0.0, lap_length=LAP_L_SHAPE):
    """MPC-CBF NLP batch: one car drawn inside +-1.5 vx of the ego; the reference's window test
    (control.py:520-522) then decides whether it enters the NLP (a car drawn across the start line does not).
    `cars` [batch, n_obs, 3] = (s0, v, ey) of the scripted cars s(t) = v t + s0 -- the raw scenario the reference-side
    fixture generator (tests/golden/tools/make_golden.py draws) replays through the reference's own code."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    j = np.arange(N + 1)

    def gen(n):
        x0 = _ego(rng, n, 0.4, 1.2)
        s_o = x0[:, 4, None] + rng.uniform(-1.5, 1.5, (n, n_obs)) * x0[:, 0, None]
        v_o = rng.uniform(0.0, 1.0, (n, n_obs))
        ey_o = _lanes(rng, (n, n_obs))
        obs_s = s_o[:, :, None] + 0.1 * j[None, None, :] * v_o[:, :, None]
        obs_ey = np.repeat(ey_o[:, :, None], N + 1, axis=2)
        return dict(x0=x0, obs_s=obs_s, obs_ey=obs_ey, cars=np.stack([s_o, v_o, ey_o], axis=2))

    p = _resample_unsafe(gen, batch, 0.2) if safe_start else gen(batch)
    p["x0"] = _lap_some(p["x0"], lapped_frac, seed)
    xt = np.tile(np.array([0.8, 0, 0, 0, 0, 0.0]), (batch, 1))
    p = _through_window(p, n_obs, lap_length)
    p.update(xt=xt, N=N, alpha=0.8, margin=0.2, lap_length=lap_length)
    return p


def cfg4_tracking_cbf(batch=16384, N=20, seed=4, n_obs=3, safe_start=True, lapped_frac=0.0, lap_length=LAP_L_SHAPE):
    """mpc_multi_agents-form NLP (control.py:251-473): per-stage ey target, alpha 0.6, margin 0.15,
    three cars ahead in the planner's front window (planner_helper.py:231-236); the controller's own +-2 vx window
    (control.py:293-309) then keeps the ones that become obstacles.  `traj` [batch, N+1, 6] is the planner-output-like
    target trajectory handed to the controller (`target_traj_xcurv`): nodes at s0 + 0.1 j vx0, so that the reference's
    clipped interpolation (control.py:373-382) returns the node values."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    j = np.arange(N + 1)

    def gen(n):
        x0 = _ego(rng, n, 0.8, 1.5)
        v_o = rng.uniform(0.5, 1.2, (n, n_obs))
        dv = np.abs(x0[:, 0, None] - v_o)
        s_o = x0[:, 4, None] + rng.uniform(0.0, 1.0, (n, n_obs)) * (4.5 * 0.4 + 0.5 * dv)
        ey_o = _lanes(rng, (n, n_obs))
        obs_s = s_o[:, :, None] + 0.1 * j[None, None, :] * v_o[:, :, None]
        obs_ey = np.repeat(ey_o[:, :, None], N + 1, axis=2)
        return dict(x0=x0, obs_s=obs_s, obs_ey=obs_ey, cars=np.stack([s_o, v_o, ey_o], axis=2))

    p = _resample_unsafe(gen, batch, 0.15) if safe_start else gen(batch)
    p["x0"] = _lap_some(p["x0"], lapped_frac, seed)
    x0 = p["x0"]
    # target: smooth lateral move from the current ey to a random lane over the horizon
    ey_goal = _lanes(rng, (batch,))
    w = (j / N)[None, :]
    blend = 3 * w ** 2 - 2 * w ** 3
    traj = np.zeros((batch, N + 1, 6))
    traj[:, :, 0] = x0[:, 0, None]
    traj[:, :, 4] = x0[:, 4, None] + 0.1 * j[None, :] * x0[:, 0, None]
    traj[:, :, 5] = x0[:, 5, None] * (1 - blend) + ey_goal[:, None] * blend
    from . import hostprep

    xt = hostprep.tracking_targets_batch(x0, traj, N)
    p = _through_window(p, n_obs, lap_length)
    p.update(xt=xt, traj=traj, N=N, alpha=0.6, margin=0.15, lap_length=lap_length)
    return p


def partial_sort_order(ey):
    """planner_helper.sort_by_ey (the reference's partial sort, overtake_traj_planner.py:70-76, quirk Q3) for a whole
    batch: ey [n, V] in iteration order -> order [n, V] (indices into the iteration order)."""
    n, V = ey.shape
    order = np.zeros((n, V), dtype=np.int64)
    rows = np.arange(n)
    for j in range(1, V):
        front = ey[rows, j] >= ey[rows, order[:, 0]]
        shifted = np.concatenate([np.full((n, 1), j), order[:, :-1]], axis=1)
        appended = order.copy()
        appended[:, j] = j
        order = np.where(front[:, None], shifted, appended)
    return order


def cfg3_raw(n_scen=1024, N=12, seed=3, V=3, track_width=1.0, lap_length=LAP_L_SHAPE):
    """Raw overtake-planner scenarios, i.e. what OvertakeTrajPlanner.get_local_traj has in hand before
    its host prep (overtake_traj_planner.py:66-92): ego + V surrounding vehicles in the front interest
    window (planner_helper.py:231-236), constant-speed predictions.  veh_info rows (s, max ey, min ey)
    are in ITERATION order (quirk Q4), predictions in the reference's partial ey order (quirk Q3)."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    opt = np.genfromtxt(os.path.join(_ROOT, "data/optimal_traj/xcurv_l_shape.csv"), delimiter=",")
    j = np.arange(N + 1)
    R = V + 1
    x = _ego(rng, n_scen, 0.8, 1.5)
    x[:, 4] = rng.uniform(0.5, 13.0, n_scen)  # keep s + look-ahead inside one lap of the optimal-trajectory table
    v_o = rng.uniform(0.5, 1.2, (n_scen, V))
    dv = np.abs(x[:, 0, None] - v_o)
    s_o = x[:, 4, None] + rng.uniform(0.0, 1.0, (n_scen, V)) * (4.5 * 0.4 + 0.5 * dv)
    ey_o = _lanes(rng, (n_scen, V))
    order = partial_sort_order(ey_o)
    rows = np.arange(n_scen)[:, None]
    obs_s = s_o[rows, order][:, :, None] + 0.1 * j[None, None, :] * v_o[rows, order][:, :, None]
    obs_ey = np.repeat(ey_o[rows, order][:, :, None], N + 1, axis=2)
    return dict(
        x=x, veh_info=np.stack([s_o, ey_o, ey_o], axis=2), cars=np.stack([s_o, v_o, ey_o], axis=2), max_dv=dv.max(axis=1),
        obs_s=obs_s, obs_ey=obs_ey,
        n_veh=np.full(n_scen, V, dtype=np.int32), opt_s=np.ascontiguousarray(opt[:, 4]),
        opt_ey=np.ascontiguousarray(opt[:, 5]), opt=opt, N=N, V=V, n_scen=n_scen, track_width=track_width,
        lap_length=lap_length, old_flag=rng.integers(-1, R, n_scen).astype(np.int32),
    )


def cfg3_planner(n_scen=1024, N=12, seed=3, V=3, track_width=1.0, lap_length=LAP_L_SHAPE):
    """cfg3_raw + the HOST prep of the mirror (planner_helper / hostprep): the crx_planner_solve arrays
    with batch = n_scen*(V+1) (scenario-major, region minor) plus the per-scenario selection arrays
    (overtake_traj_planner.py:182-197, :205-246).  crx_planner_prep produces the same arrays on the device."""
    from planning import planner_helper as ph

    from . import hostprep

    w = cfg3_raw(n_scen, N, seed, V, track_width, lap_length)
    R = V + 1
    x = w["x"]
    bez = np.zeros((n_scen, R, N + 1, 2))
    for i in range(n_scen):
        cp = ph.bezier_control_points(V, w["veh_info"][i], w["max_dv"][i], 0.5, track_width, lap_length, 0.2, w["opt"], x[i])
        bez[i] = ph.bezier_polylines(cp, N)
    lb, ub = hostprep.planner_ey_bounds(x, w["obs_s"], w["obs_ey"], w["n_veh"], track_width, lap_length, N)
    return dict(
        x0=np.repeat(x, R, axis=0), bez_s=bez[..., 0].reshape(n_scen * R, N + 1),
        bez_ey=bez[..., 1].reshape(n_scen * R, N + 1), ey_lb=lb.reshape(n_scen * R, N),
        ey_ub=ub.reshape(n_scen * R), N=N, V=V, n_scen=n_scen, n_veh=w["n_veh"], obs_s=w["obs_s"], obs_ey=w["obs_ey"],
        old_flag=w["old_flag"], lap_length=lap_length, raw=w,
    )


def multi_tests_traffic(n, num_veh=3, seed=0):
    """The scripted cars of the reference's Monte-Carlo experiment (car_racing/tests/overtake_planner_test.py:81-90,
    `--multi-tests` / `--random-other-agents`): car i drives s(t) = 0.1 randint(0, 10) t + 3 + randint(0, 14) at the constant
    offset ey = 0.7 - 0.1 randint(0, 14) (Python's randint is inclusive).  Returns s0, v, ey as [n, num_veh] arrays."""
    rng = np.random.default_rng(seed)
    v = 0.1 * rng.integers(0, 11, (n, num_veh))
    s0 = 3.0 + rng.integers(0, 15, (n, num_veh))
    ey = 0.7 - 0.1 * rng.integers(0, 15, (n, num_veh))
    return s0.astype(float), v.astype(float), ey.astype(float)
