"""Generate tests/golden/*.npz by running the reference's OWN, unmodified solver front-ends
(/root/reference/car_racing) against the recording CasADi stand-in (shim/casadi.py).

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/tools/make_golden.py            # writes tests/golden/*.npz

What a fixture holds (all data, no reference source):
  * the raw scenario the reference was called with (ego state, vehicle states, obstacle predictions
    as returned by the reference's vehicle models, Bezier polylines as computed by the reference's
    planner_helper, parameters),
  * what the reference returned (u, x_pred, direction_flag, trajectories, fall-back use),
  * the solution of the recorded problem with an explicit, solver-agnostic KKT certificate
    (stationarity / feasibility / multiplier sign / complementarity evaluated on the recorded
    expression graphs).
IPOPT never ran (not installable offline): the per-problem solver is tools/ipm_dense.py
(+ exact active-set polish for the QPs, + a HiGHS feasibility verdict for the QPs).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.normpath(os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

M = ref_harness.install()
import sympy as sp  # noqa: E402
from scipy.optimize import linprog  # noqa: E402

import ipm_dense  # noqa: E402
import nlp_solve  # noqa: E402

base, offboard, racing_env = M["base"], M["offboard"], M["racing_env"]
control, casadi, planner_mod = M["control"], M["casadi"], M["planner"]

RECORDS = []


def golden_solver(opti):
    """Opti.solve() stand-in: certified solve of the recorded problem."""
    n = opti.nvar
    rng = np.random.default_rng(0)
    affine = nlp_solve.is_affine(opti, rng)
    info = dict(affine=bool(affine), success=False, reason="")
    f0, g0, ce0, Je0, ci0, Ji0 = opti.eval_all(np.zeros(n))
    if affine:
        res = linprog(np.zeros(n), A_ub=-Ji0, b_ub=ci0, A_eq=Je0, b_eq=-ce0,
                      bounds=[(None, None)] * n, method="highs")
        info["lp_status"] = int(res.status)
        if res.status == 2:
            info["reason"] = "infeasible (HiGHS)"
            z = np.zeros(n)
            if ELASTIC_ON_INFEASIBLE[0]:
                z = elastic_qp(opti, ELASTIC_ON_INFEASIBLE[0])
                info["reason"] += "; returned the exact-L1-penalty minimiser as the 'non-converged iterate'"
            RECORDS.append((opti, z, info))
            return z, info
    gopts = ipm_dense.Opts()
    gopts.tol = 1e-11  # goldens are solved two orders tighter than the product default (1e-8)
    r = ipm_dense.solve_recorded(opti, gopts)
    z, nu = r["z"], r["nu_full"]
    info["ipm_status"] = int(r["status"])
    info["ipm_iters"] = int(r["iters"])
    info["const_violation"] = float(r["const_violation"])
    if affine and r["status"] == 0:
        H = np.zeros((n, n))
        for i in range(n):
            e = np.zeros(n)
            e[i] = 1.0
            H[:, i] = opti.eval_all(e)[1] - g0
        H = 0.5 * (H + H.T)
        zp, ok = nlp_solve._polish_qp(H, g0, Je0, -ce0, Ji0, -ci0, z)
        info["polished"] = bool(ok)
        if ok:
            z, nu = zp, None
    cert = nlp_solve.kkt_certificate(opti, z, nu)
    info.update(cert)
    gscale = max(1.0, float(np.abs(opti.eval_all(z)[1]).max()))
    info["success"] = bool(
        r["status"] == 0
        and r["const_violation"] <= 1e-8
        and cert["stationarity"] <= 1e-7 * gscale
        and cert["eq_violation"] <= 1e-9
        and cert["ineq_violation"] <= 1e-8
        and cert["min_multiplier"] >= -1e-7
    )
    if not info["success"]:
        info["reason"] = "status %d const_violation %.2e stationarity %.2e" % (
            r["status"], r["const_violation"], cert["stationarity"])
    RECORDS.append((opti, z, info))
    return z, info


# What IPOPT leaves in opti.debug after "Converged to a point of local infeasibility" is the state
# of its restoration phase and cannot be restated.  Where a closed-loop generator has to continue
# past an infeasible QP (the LMPC laps: the reference pins its terminal slack to zero,
# control.py:689-690, so its own QP is infeasible whenever the learned model cannot reach the safe
# set), the stand-in below returns the minimiser of  cost + rho * ||equality violation||_1  subject to
# the inequalities.  Such steps are flagged (success = False) and no parity claim is made on them.
ELASTIC_ON_INFEASIBLE = [0.0]


def elastic_qp(opti, rho):
    n = opti.nvar
    f0, g0, ce0, Je, ci0, Ji = opti.eval_all(np.zeros(n))
    H = np.zeros((n, n))
    for i in range(n):
        e = np.zeros(n)
        e[i] = 1.0
        H[:, i] = opti.eval_all(e)[1] - g0
    H = 0.5 * (H + H.T)
    me = Je.shape[0]
    # variables w = (z, p, q):  Je z + ce0 + p - q = 0,  p, q >= 0
    Ae = np.hstack([Je, np.eye(me), -np.eye(me)])
    nw = n + 2 * me
    _, S, Vt = np.linalg.svd(Ae)
    Z = Vt[me:].T
    wp = np.linalg.lstsq(Ae, -ce0, rcond=None)[0]
    Jw = np.zeros((len(ci0) + 2 * me, nw))
    Jw[: len(ci0), :n] = Ji
    Jw[len(ci0):, n:] = np.eye(2 * me)
    c0 = np.concatenate([ci0, np.zeros(2 * me)])
    Hw = np.zeros((nw, nw))
    Hw[:n, :n] = H
    gw = np.concatenate([g0, rho * np.ones(2 * me)])
    Hr, JZ = Z.T @ Hw @ Z, Jw @ Z

    def fun(v):
        w = wp + Z @ v
        return 0.5 * w @ Hw @ w + gw @ w, Z.T @ (Hw @ w + gw), c0 + Jw @ w, JZ

    o = ipm_dense.Opts()
    o.tol = 1e-9
    r = ipm_dense.solve(fun, lambda v, nu: Hr, np.zeros(Z.shape[1]), opts=o)
    return (wp + Z @ r["v"])[:n]


casadi.Opti.solver_fn = staticmethod(golden_solver)


# in-process stand-ins for the fork fan-out (overtake_traj_planner.py:177-197) so that the
# per-region records reach this process
class _Proc:
    def __init__(self, target, args):
        self.target, self.args = target, args

    def start(self):
        self.target(*self.args)

    def join(self):
        pass


class _Mgr:
    def dict(self):
        return {}


planner_mod.Process = _Proc
planner_mod.Manager = _Mgr

TRACK_SPEC = np.genfromtxt("data/track_layout/l_shape.csv", delimiter=",")
OPTI_XCURV = np.genfromtxt("data/optimal_traj/xcurv_l_shape.csv", delimiter=",")
T = sp.symbols("t")


def make_track(width=1.0):
    return racing_env.ClosedTrack(TRACK_SPEC, track_width=width)


def cert_fields(info):
    keys = ("f", "stationarity", "eq_violation", "ineq_violation", "min_multiplier", "complementarity")
    return np.array([info.get(k, np.nan) for k in keys], dtype=float)


# -------------------------------------------------------------------------------------------------
# mpccbf (control.py:476-607)
# -------------------------------------------------------------------------------------------------
def mpccbf_case(x0, cars, N=10, alpha=0.8, vt=0.8, width=1.0, time=0.0, car_dims=None):
    """cars: list of (s0, v, ey) for NoDynamicsModel obstacles s(t) = v t + s0, ey(t) = ey; car_dims: (length, width) per car
    (default CarParam(): 0.4 x 0.2)."""
    track = make_track(width)
    ego = offboard.DynamicBicycleModel(name="ego", param=base.CarParam(), system_param=base.SystemParam())
    ego.set_zero_noise()
    par = base.MPCCBFRacingParam(vt=vt, num_horizon=N, alpha=alpha)
    ego.set_state_curvilinear(np.array(x0, float))
    ego.set_state_global(np.zeros(6))
    ego.set_ctrl_policy(offboard.MPCCBFRacing(par, ego.system_param))
    ego.ctrl_policy.set_timestep(0.1)
    ego.set_track(track)
    ego.ctrl_policy.set_track(track)
    sim = offboard.CarRacingSim()
    sim.set_timestep(0.1)
    sim.set_track(track)
    sim.add_vehicle(ego)
    ego.ctrl_policy.set_racing_sim(sim)
    for i, (s0, v, ey) in enumerate(cars):
        c = offboard.NoDynamicsModel(name="car%d" % (i + 1), param=base.CarParam() if car_dims is None else base.CarParam(length=car_dims[i][0], width=car_dims[i][1]))
        c.set_track(track)
        c.set_state_curvilinear_func(T, v * T + s0, ey + 0.0 * T)
        sim.add_vehicle(c)
        c.time = time
    ego.ctrl_policy.time = time
    del RECORDS[:]
    ego.ctrl_policy.set_state(ego.xcurv, ego.xglob)
    ego.ctrl_policy.calc_input()
    opti, z, info = RECORDS[-1]
    n_obs_rec = (opti.nvar - (8 * N + 6)) // (N + 1)
    # obstacle predictions exactly as the reference's vehicle model returns them
    preds = []
    for i in range(len(cars)):
        tr, _ = sim.vehicles["car%d" % (i + 1)].get_trajectory_nsteps(time, 0.1, N + 1)
        preds.append(tr)
    X = z[: 6 * (N + 1)].reshape(N + 1, 6)
    U = z[6 * (N + 1): 6 * (N + 1) + 2 * N].reshape(N, 2)
    S = z[6 * (N + 1) + 2 * N:].reshape(N + 1, n_obs_rec).T if n_obs_rec else np.zeros((0, N + 1))
    return dict(
        x0=np.array(x0, float), N=N, alpha=alpha, vt=vt, width=width, lap_length=track.lap_length,
        cars=np.array(cars, float).reshape(-1, 3), obs_pred=np.array(preds).reshape(len(cars), 6, N + 1),
        n_obs_in_problem=n_obs_rec, u_returned=np.array(ego.ctrl_policy.u, float),
        X=X, U=U, sigma=S, success=info["success"], cert=cert_fields(info),
        ipm_iters=info.get("ipm_iters", -1), const_violation=info.get("const_violation", 0.0),
    )


def gen_mpccbf():
    cases = {
        "free_road": dict(x0=[0.7, 0.01, 0.02, 0.03, 2.9, 0.05], cars=[(4.0, 0.2, 0.1)]),
        "blocked_brake": dict(x0=[0.7, 0.01, 0.02, 0.03, 2.9, 0.05], cars=[(3.6, 0.2, 0.05)]),
        "pass_left": dict(x0=[1.0, 0.0, 0.0, 0.0, 3.0, 0.35], cars=[(3.9, 0.3, -0.1)]),
        "pass_right": dict(x0=[1.0, 0.0, 0.0, 0.0, 3.0, -0.4], cars=[(3.8, 0.3, 0.1)]),
        "two_cars": dict(x0=[0.9, 0.0, 0.0, 0.0, 6.0, 0.0], cars=[(6.9, 0.4, 0.3), (7.4, 0.2, -0.35)]),
        "out_of_window": dict(x0=[0.8, 0.0, 0.0, 0.0, 1.0, 0.0], cars=[(8.0, 0.2, 0.1)]),
        "behind_ego": dict(x0=[0.8, 0.0, 0.0, 0.0, 5.0, 0.1], cars=[(4.3, 0.9, 0.15)]),
        "n12_alpha06": dict(x0=[1.1, 0.02, -0.05, 0.04, 9.0, -0.2], cars=[(9.9, 0.5, 0.0)], N=12, alpha=0.6),
        "lap_quirk_q1": dict(x0=[0.8, 0.0, 0.0, 0.0, 19.6, 0.0], cars=[(1.2, 0.2, 0.4)]),
        "x0_off_track_q9": dict(x0=[0.8, 0.0, 0.0, 0.05, 3.0, 1.05], cars=[(4.0, 0.2, 0.1)]),
        "late_time_q6": dict(x0=[0.9, 0.0, 0.0, 0.0, 4.6, -0.1], cars=[(4.0, 0.2, 0.1)], time=6.0),
    }
    out = {}
    for name, kw in cases.items():
        r = mpccbf_case(**kw)
        print("mpccbf %-18s n_obs %d success %s f %.8f stat %.1e iters %d u %s" % (
            name, r["n_obs_in_problem"], r["success"], r["cert"][0], r["cert"][1], r["ipm_iters"], r["u_returned"]))
        for k, v in r.items():
            out[name + "/" + k] = v
    out["names"] = np.array(sorted(cases))
    np.savez_compressed(os.path.join(OUT, "mpccbf.npz"), **out)


# -------------------------------------------------------------------------------------------------
# planner (overtake_traj_planner.py:44-379) and mpc_multi_agents (control.py:251-473)
# -------------------------------------------------------------------------------------------------
def planner_case(x0, cars, dyn_cars=(), N=10, old_flag=None, width=1.0, time=0.0, raw_s=None):
    """cars: NoDynamics obstacles (s0, v, ey); dyn_cars: DynamicBicycleModel obstacles given by xcurv."""
    track = make_track(width)
    par = base.RacingGameParam(timestep=0.1, num_horizon_planner=N, num_horizon_ctrl=N)
    ego = offboard.DynamicBicycleModel(name="ego", param=base.CarParam(), system_param=base.SystemParam())
    xraw = np.array(x0, float)
    if raw_s is not None:
        xraw[4] = raw_s
    ego.set_state_curvilinear(xraw)
    ego.set_state_global(np.zeros(6))
    ego.set_track(track)
    ego.set_timestep(0.1)
    vehicles = {"ego": ego}
    for i, (s0, v, ey) in enumerate(cars):
        c = offboard.NoDynamicsModel(name="car%d" % (i + 1), param=base.CarParam())
        c.set_track(track)
        c.set_timestep(0.1)
        c.set_state_curvilinear_func(T, v * T + s0, ey + 0.0 * T)
        c.time = time
        c.xcurv, c.xglob = c.get_estimation(time)
        vehicles[c.name] = c
    for i, xc in enumerate(dyn_cars):
        c = offboard.DynamicBicycleModel(name="dyn%d" % (i + 1), param=base.CarParam(), system_param=base.SystemParam())
        c.set_track(track)
        c.set_timestep(0.1)
        c.set_state_curvilinear(np.array(xc, float))
        X, Y = track.get_global_position(xc[4], xc[5])
        psi = track.get_orientation(xc[4], xc[5])
        c.set_state_global(np.array([xc[0], xc[1], xc[2], psi + xc[3], X, Y]))
        vehicles[c.name] = c
    pl = planner_mod.OvertakeTrajPlanner(par)
    pl.vehicles, pl.agent_name, pl.track, pl.opti_traj_xcurv = vehicles, "ego", track, OPTI_XCURV
    x = np.array(x0, float)  # start-line-wrapped copy handed to the planner (utils/base.py:460-462)
    flag, interest = pl.get_overtake_flag(x)
    res = dict(x_wrapped=x, x_raw=xraw, N=N, width=width, lap_length=track.lap_length,
               overtake_flag=bool(flag), interest=np.array(sorted(interest)),
               old_flag=-1 if old_flag is None else int(old_flag))
    names = [n for n in vehicles if n != "ego"]
    res["veh_names"] = np.array(names)
    res["veh_xcurv"] = np.array([vehicles[n].xcurv for n in names]).reshape(len(names), 6)
    res["veh_is_interest"] = np.array([n in interest for n in names])
    if not flag:
        return res
    del RECORDS[:]
    (traj, traj_glob, dflag, sorted_veh, bez_glob, solve_time, all_bez_glob, all_traj_glob) = pl.get_local_traj(
        x, time, interest, None, None, None, None, old_flag)
    V = len(sorted_veh)
    recs = list(RECORDS)
    assert len(recs) == V + 1
    res.update(
        sorted_vehicles=np.array(sorted_veh),
        obs_pred=np.array([pl.obs_infos[n] for n in sorted_veh]).reshape(V, 6, N + 1),
        bezier_xcurvs=pl.bezier_xcurvs.copy(),
        direction_flag=int(dflag), traj_xcurv=np.array(traj), traj_xglob=np.array(traj_glob),
        region_success=np.array([r[2]["success"] for r in recs]),
        region_cert=np.array([cert_fields(r[2]) for r in recs]),
        region_X=np.array([pl_sol for pl_sol in all_local_xcurv(pl, recs, N)]),
        region_lp_status=np.array([r[2].get("lp_status", -1) for r in recs]),
        all_traj_xglob=np.array(all_traj_glob), all_bezier_xglob=np.array(all_bez_glob),
    )
    # follow-up tracking controller (utils/base.py:558-572).  With a DynamicBicycleModel obstacle
    # the reference itself raises TypeError (control.py:296 always uses the 3-argument form).
    res["mma_present"] = not dyn_cars
    if dyn_cars:
        return res
    del RECORDS[:]
    u, x_pred = control.mpc_multi_agents(
        x, par, track, None, None, None, base.SystemParam(), target_traj_xcurv=traj, vehicles=vehicles,
        agent_name="ego", direction_flag=dflag, target_traj_xglob=traj_glob, sorted_vehicles=sorted_veh)
    opti, z, info = RECORDS[-1]
    n_obs_rec = (opti.nvar - (8 * N + 6)) // (N + 1)
    res.update(
        mma_u=np.array(u, float), mma_x_pred=np.array(x_pred, float), mma_success=info["success"],
        mma_cert=cert_fields(info), mma_n_obs=n_obs_rec,
        mma_X=z[: 6 * (N + 1)].reshape(N + 1, 6), mma_U=z[6 * (N + 1): 8 * N + 6].reshape(N, 2),
        mma_sigma=z[8 * N + 6:].reshape(N + 1, n_obs_rec).T if n_obs_rec else np.zeros((0, N + 1)),
        mma_obs_pred=np.array([vehicles[n].get_trajectory_nsteps(None, 0.1, N + 1)[0] if vehicles[n].no_dynamics
                               else vehicles[n].get_trajectory_nsteps(N + 1)[0] for n in sorted_veh]).reshape(V, 6, N + 1),
    )
    return res


def all_local_xcurv(pl, recs, N):
    """Per-region trajectories as the reference stored them (solution or fall-back), (V+1, N+1, 6)."""
    # solve_optimization_problem returns solution_xvar as 4th value; re-derive from the records the
    # same way the reference does: success -> sol.value(opti_xvar); failure -> fall-back (:365-374)
    out = []
    for idx, (opti, z, info) in enumerate(recs):
        if info["success"]:
            out.append(z[: 6 * (N + 1)].reshape(N + 1, 6))
        else:
            xc = pl.xcurv_ego
            sol = np.zeros((N + 1, 6))
            for j in range(N + 1):
                stmp = xc[4] + 1.1 * j * 0.1 * xc[0]
                sol[j, 0] = 1.1 * xc[0]
                sol[j, 4] = stmp
                stmp = np.clip(stmp, pl.bezier_xcurvs[idx, 0, 0], pl.bezier_xcurvs[idx, -1, 0])
                sol[j, 5] = pl.bezier_funcs[idx](stmp)
            out.append(sol)
    return out


def gen_planner():
    cases = {
        "one_car_ahead": dict(x0=[1.2, 0.0, 0.0, 0.0, 5.0, 0.1], cars=[(6.0, 0.7, -0.1)]),
        "one_car_left": dict(x0=[1.3, 0.0, 0.0, 0.02, 8.0, -0.2], cars=[(8.9, 0.6, 0.4)]),
        "two_cars": dict(x0=[1.3, 0.0, 0.0, 0.0, 5.0, 0.0], cars=[(5.9, 0.7, -0.5), (6.4, 0.72, -0.2)]),
        "three_cars": dict(x0=[1.4, 0.01, 0.0, 0.0, 10.0, 0.1], cars=[(10.7, 0.7, -0.5), (11.2, 0.72, 0.0), (11.6, 0.74, 0.5)]),
        "three_cars_oldflag": dict(x0=[1.4, 0.01, 0.0, 0.0, 10.0, 0.1], cars=[(10.7, 0.7, -0.5), (11.2, 0.72, 0.0), (11.6, 0.74, 0.5)], old_flag=0),
        "alongside": dict(x0=[1.0, 0.0, 0.0, 0.0, 12.0, 0.3], cars=[(12.2, 0.9, -0.2)]),
        "n12": dict(x0=[1.2, 0.0, 0.0, 0.0, 3.0, 0.0], cars=[(4.2, 0.6, 0.2)], N=12),
        "dyn_obstacle": dict(x0=[1.2, 0.0, 0.0, 0.0, 14.0, 0.0], cars=[], dyn_cars=[[0.6, 0.0, 0.0, 0.0, 15.0, 0.2]]),
        "start_line_q5": dict(x0=[1.2, 0.0, 0.0, 0.0, 0.2, 0.0], cars=[(1.3, 0.6, 0.2)], raw_s=0.2 + 19.22957795362994),
        "no_interest": dict(x0=[1.0, 0.0, 0.0, 0.0, 2.0, 0.0], cars=[(9.0, 0.7, 0.0)]),
    }
    out = {}
    for name, kw in cases.items():
        r = planner_case(**kw)
        if r["overtake_flag"]:
            print("planner %-18s V %d flag %d region_success %s lp %s mma_success %s n_obs %s u %s" % (
                name, len(r["sorted_vehicles"]), r["direction_flag"], r["region_success"],
                r["region_lp_status"], r.get("mma_success"), r.get("mma_n_obs"), r.get("mma_u")))
        else:
            print("planner %-18s no vehicle of interest" % name)
        for k, v in r.items():
            out[name + "/" + k] = v
    out["names"] = np.array(sorted(cases))
    np.savez_compressed(os.path.join(OUT, "planner.npz"), **out)


def path_case(x0, cars, N=10, alpha=0.8, width=1.0, time=0.0):
    """OvertakePathPlanner.get_local_path (planning/overtake_path_planner.py:37-183) on scripted cars."""
    from planning import overtake_path_planner as opp

    track = make_track(width)
    par = base.RacingGameParam(timestep=0.1, num_horizon_planner=N, num_horizon_ctrl=N, alpha=alpha)
    ego = offboard.DynamicBicycleModel(name="ego", param=base.CarParam(), system_param=base.SystemParam())
    ego.set_state_curvilinear(np.array(x0, float)); ego.set_state_global(np.zeros(6)); ego.set_track(track); ego.set_timestep(0.1)
    vehicles = {"ego": ego}
    for i, (s0, v, ey) in enumerate(cars):
        c = offboard.NoDynamicsModel(name="car%d" % (i + 1), param=base.CarParam())
        c.set_track(track); c.set_timestep(0.1)
        c.set_state_curvilinear_func(T, v * T + s0, ey + 0.0 * T)
        c.time = time
        c.xcurv, c.xglob = c.get_estimation(time)
        vehicles[c.name] = c
    pl = opp.OvertakePathPlanner(par)
    pl.vehicles, pl.agent_name, pl.track, pl.opti_traj_xcurv = vehicles, "ego", track, OPTI_XCURV
    x = np.array(x0, float)
    flag, interest = pl.get_overtake_flag(x)
    names = [n for n in vehicles if n != "ego"]
    res = dict(x=x, N=N, alpha=alpha, width=width, lap_length=track.lap_length, overtake_flag=bool(flag), time=time,
               cars=np.array(cars, float).reshape(-1, 3), veh_names=np.array(names),
               veh_xcurv=np.array([vehicles[n].xcurv for n in names]).reshape(len(names), 6),
               veh_is_interest=np.array([n in interest for n in names]))
    if not flag:
        return res
    del RECORDS[:]
    (traj, traj_glob, dflag, sorted_veh, bez_glob, solve_time, all_bez_glob, all_traj_glob) = pl.get_local_path(x, time, interest)
    recs = list(RECORDS)
    V = len(sorted_veh)
    assert len(recs) == V + 1
    if os.environ.get("CRX_GOLDEN_DEBUG"):
        for r in recs:
            print("   region:", {k: v for k, v in r[2].items() if k in ("success", "reason", "lp_status", "ipm_status", "ipm_iters", "stationarity", "ineq_violation", "eq_violation", "min_multiplier", "const_violation", "polished")})
    res.update(sorted_vehicles=np.array(sorted_veh), direction_flag=int(dflag), traj_xcurv=np.array(traj, float),
               traj_xglob=np.array(traj_glob, float), region_success=np.array([r[2]["success"] for r in recs]),
               region_E=np.array([r[1][:N + 1] for r in recs]), region_cert=np.array([cert_fields(r[2]) for r in recs]),
               region_lp_status=np.array([r[2].get("lp_status", -1) for r in recs]),
               all_bezier_xglob=np.array(all_bez_glob, float))
    return res


def gen_path():
    cases = {
        "one_car_ahead": dict(x0=[1.2, 0.0, 0.0, 0.0, 5.0, 0.1], cars=[(6.0, 0.7, -0.1)]),
        "one_car_left": dict(x0=[1.3, 0.0, 0.0, 0.02, 8.0, -0.2], cars=[(8.9, 0.6, 0.4)]),
        "two_cars": dict(x0=[1.3, 0.0, 0.0, 0.0, 5.0, 0.0], cars=[(5.9, 0.7, -0.5), (6.4, 0.72, -0.2)]),
        "three_cars": dict(x0=[1.4, 0.01, 0.0, 0.0, 10.0, 0.1], cars=[(10.7, 0.7, -0.5), (11.2, 0.72, 0.0), (11.6, 0.74, 0.5)]),
        "wide_track": dict(x0=[1.2, 0.0, 0.0, 0.0, 3.0, 0.0], cars=[(4.2, 0.6, 0.2)], width=2.0),
        "wide_two": dict(x0=[1.3, 0.0, 0.0, 0.0, 12.0, -0.3], cars=[(12.9, 0.7, 0.9), (13.3, 0.6, -0.9)], width=2.0, N=12),
        "alpha05": dict(x0=[1.0, 0.0, 0.0, 0.0, 12.0, 0.3], cars=[(12.6, 0.9, -0.2)], alpha=0.5, width=1.6),
        "no_interest": dict(x0=[1.0, 0.0, 0.0, 0.0, 2.0, 0.0], cars=[(9.0, 0.7, 0.0)]),
    }
    out = {}
    for name, kw in cases.items():
        r = path_case(**kw)
        if r["overtake_flag"]:
            print("path %-14s V %d flag %d region_success %s lp %s" % (name, len(r["sorted_vehicles"]), r["direction_flag"],
                                                                       r["region_success"], r["region_lp_status"]))
        else:
            print("path %-14s no vehicle of interest" % name)
        for k, v in r.items():
            out[name + "/" + k] = v
    out["names"] = np.array(sorted(cases))
    np.savez_compressed(os.path.join(OUT, "path_planner.npz"), **out)


def gen_harness():
    """Solver-free host code: track geometry, plant step, vehicle predictions, PID closed loop."""
    from system import vehicle_dynamics as vd

    out = {}
    rng = np.random.default_rng(7)
    for tname in ("l_shape", "m_shape", "goggle", "ellipse"):
        spec = np.genfromtxt("data/track_layout/%s.csv" % tname, delimiter=",")
        tr = racing_env.ClosedTrack(spec, track_width=1.0)
        s = np.concatenate([rng.uniform(0, tr.lap_length, 60), tr.point_and_tangent[:, 3] + 1e-9,
                            [0.0, tr.lap_length - 1e-6, tr.lap_length + 0.5, -0.3]])
        ey = rng.uniform(-0.8, 0.8, len(s))
        out[tname + "/lap_length"] = tr.lap_length
        out[tname + "/table"] = tr.point_and_tangent
        out[tname + "/s"] = s
        out[tname + "/ey"] = ey
        out[tname + "/xy"] = np.array([tr.get_global_position(a, b) for a, b in zip(s, ey)])
        out[tname + "/psi"] = np.array([tr.get_orientation(a, b) for a, b in zip(s, ey)])
        sc = rng.uniform(0, tr.lap_length, 60)  # interior points only: the reference raises at joints
        out[tname + "/s_curv"] = sc
        out[tname + "/curv"] = np.array([tr.get_curvature(a) for a in sc])
    # plant: 40 random single Euler steps
    dyn = base.CarParam().dynamics_param
    xg = rng.normal(size=(40, 6)); xc = rng.normal(size=(40, 6)) * 0.3
    xc[:, 0] = rng.uniform(0.3, 1.5, 40); xg[:, 0:3] = xc[:, 0:3]
    u = np.stack([rng.uniform(-0.5, 0.5, 40), rng.uniform(-1, 1, 40)], axis=1)
    curv = rng.uniform(-0.7, 0.7, 40)
    res = [vd.vehicle_dynamics(dyn, curv[i], xg[i], xc[i], 0.001, u[i]) for i in range(40)]
    out.update({"plant/xglob": xg, "plant/xcurv": xc, "plant/u": u, "plant/curv": curv,
                "plant/xglob_next": np.array([r[0] for r in res]), "plant/xcurv_next": np.array([r[1] for r in res])})
    # PID closed loop, zero noise, 30 steps on l_shape (ModelBase + plant + lap bookkeeping)
    track = make_track(0.8)
    ego = offboard.DynamicBicycleModel(name="ego", param=base.CarParam(), system_param=base.SystemParam())
    ego.set_zero_noise()
    ego.set_state_curvilinear(np.zeros(6)); ego.set_state_global(np.zeros(6)); ego.start_logging()
    ego.set_ctrl_policy(offboard.PIDTracking(vt=0.8)); ego.ctrl_policy.set_timestep(0.1)
    ego.set_track(track); ego.ctrl_policy.set_track(track)
    sim = offboard.CarRacingSim(); sim.set_timestep(0.1); sim.set_track(track); sim.add_vehicle(ego)
    ego.ctrl_policy.set_racing_sim(sim)
    sim.sim(sim_time=3.0)
    out["pid/xcurv_log"] = np.array(ego.xcurv_log)
    out["pid/xglob_log"] = np.array(ego.xglob_log)
    # predictions of both vehicle kinds
    car = offboard.NoDynamicsModel(name="car1", param=base.CarParam()); car.set_track(track); car.set_timestep(0.1)
    car.set_state_curvilinear_func(T, 0.2 * T + 4.0, 0.1 + 0.0 * T); car.time = 1.3
    out["pred/nodyn"] = car.get_trajectory_nsteps(0.0, 0.1, 11)[0]
    dynv = offboard.DynamicBicycleModel(name="d", param=base.CarParam(), system_param=base.SystemParam())
    dynv.set_track(track); dynv.set_timestep(0.1)
    dynv.set_state_curvilinear(np.array([0.9, 0.02, 0.1, 0.05, 18.9, 0.2])); dynv.set_state_global(np.array([0.9, 0.02, 0.1, 0.3, 1.0, 0.5]))
    out["pred/dyn"] = dynv.get_trajectory_nsteps(11)[0]
    # interest test grid (planner_helper.check_ego_agent_distance)
    par = base.RacingGameParam(timestep=0.1)
    class V:  # noqa: E306
        def __init__(self, xc): self.xcurv, self.param = np.array(xc, float), base.CarParam()
    cases = []
    for se in (0.5, 5.0, 18.9, 19.5):
        for sa in (0.2, 1.0, 5.3, 6.5, 7.5, 18.8, 19.1, 20.0):
            for dv in (0.0, 0.6):
                e, a = V([1.0, 0, 0, 0, se, 0]), V([1.0 - dv, 0, 0, 0, sa, 0])
                cases.append((se, sa, dv, float(M["planner_helper"].check_ego_agent_distance(e, a, par, track.lap_length))))
    out["interest/cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "harness.npz"), **out)
    print("harness fixture written:", len(out), "arrays")


def gen_closed_loop(steps=int(os.environ.get("CRX_GOLDEN_STEPS", "150")), layout=os.environ.get("CRX_GOLDEN_TRACK", "l_shape")):
    """The scenario of the reference's tests/auto_mpccbf_test.py:9-46 (zero noise), first `steps`
    control steps, every NLP solved by the certified golden solver.  CRX_GOLDEN_TRACK selects another
    layout of data/track_layout (car_racing/tests/mpccbf_test.py --track-layout)."""
    track = (make_track(1.0) if layout == "l_shape" else
             racing_env.ClosedTrack(np.genfromtxt("data/track_layout/%s.csv" % layout, delimiter=","), track_width=1.0))
    ego = offboard.DynamicBicycleModel(name="ego", param=base.CarParam(edgecolor="black"), system_param=base.SystemParam())
    ego.set_zero_noise()
    ego.set_state_curvilinear(np.zeros((6,))); ego.set_state_global(np.zeros((6,))); ego.start_logging()
    ego.set_ctrl_policy(offboard.MPCCBFRacing(base.MPCCBFRacingParam(vt=0.8), ego.system_param))
    ego.ctrl_policy.set_timestep(0.1); ego.set_track(track); ego.ctrl_policy.set_track(track)
    car1 = offboard.NoDynamicsModel(name="car1", param=base.CarParam()); car1.set_track(track)
    car1.set_state_curvilinear_func(T, 0.2 * T + 4.0, 0.1 + 0.0 * T); car1.start_logging()
    car2 = offboard.NoDynamicsModel(name="car2", param=base.CarParam()); car2.set_track(track)
    car2.set_state_curvilinear_func(T, 0.2 * T + 10.0, -0.1 + 0.0 * T); car2.start_logging()
    sim = offboard.CarRacingSim(); sim.set_timestep(0.1); sim.set_track(track)
    sim.add_vehicle(ego); ego.ctrl_policy.set_racing_sim(sim); sim.add_vehicle(car1); sim.add_vehicle(car2)
    del RECORDS[:]
    sim.sim(sim_time=steps * 0.1)
    ok = np.array([r[2]["success"] for r in RECORDS])
    nobs = np.array([(r[0].nvar - 86) // 11 for r in RECORDS])
    np.savez_compressed(os.path.join(OUT, "closed_loop_mpccbf.npz" if layout == "l_shape" else "closed_loop_mpccbf_%s.npz" % layout), steps=steps,
                        ego_xcurv=np.array(ego.xcurv_log), ego_xglob=np.array(ego.xglob_log),
                        ego_u=np.array([r[1][66:68] for r in RECORDS]), solve_success=ok, n_obs=nobs,
                        car1_xcurv=np.array(car1.xcurv_log), car2_xcurv=np.array(car2.xcurv_log))
    print("closed loop: %d steps, %d certified solves, obstacles per step max %d, final ego s %.3f" % (
        steps, ok.sum(), nobs.max(), ego.xcurv[4]))


def gen_racing_game(laps=int(os.environ.get("CRX_GOLDEN_LAPS", "4")),
                    keep_every=int(os.environ.get("CRX_GOLDEN_LMPC_EVERY", "8"))):
    """The scenario of the reference's tests/auto_racing_game_test.py:11-113 (zero noise) without its
    plotting tail: lap 0 PID, lap 1 mpc-lti, lap 2 LMPC, lap 3 LMPC + overtaking of two cars.
    Stores the closed-loop logs, and for every `keep_every`-th control.lmpc call the complete
    problem data (LTV model from the reference's regression, safe-set points, Q-function, u_old) and
    the certified solution."""
    from control import lmpc_helper

    track = make_track(1.0)
    opti_xglob = np.genfromtxt("data/optimal_traj/xglob_l_shape.csv", delimiter=",")
    timestep = 0.1
    ego = offboard.DynamicBicycleModel(name="ego", param=base.CarParam(edgecolor="black"), system_param=base.SystemParam())
    ego.set_timestep(timestep)
    pid = offboard.PIDTracking(vt=0.7, eyt=0.0); pid.set_timestep(timestep)
    ego.set_ctrl_policy(pid); pid.set_track(track)
    ego.set_state_curvilinear(np.zeros((6,))); ego.set_state_global(np.zeros((6,))); ego.start_logging(); ego.set_track(track)
    mpc_lti = offboard.MPCTracking(base.MPCTrackingParam(vt=0.7, eyt=0.0), ego.system_param)
    mpc_lti.set_timestep(timestep); mpc_lti.set_track(track)
    ego.set_zero_noise()
    time_lmpc = 10000 * timestep
    lmpc_param = base.LMPCRacingParam(timestep=timestep, lap_number=laps, time_lmpc=time_lmpc)
    game_param = base.RacingGameParam(timestep=timestep, alpha=0.8, num_horizon_planner=10)
    lmpc_ctrl = offboard.LMPCRacingGame(lmpc_param, racing_game_param=game_param, system_param=ego.system_param)
    lmpc_ctrl.set_track(track); lmpc_ctrl.set_timestep(timestep); lmpc_ctrl.set_opti_traj(OPTI_XCURV, opti_xglob)
    lmpc_ctrl.openloop_prediction = lmpc_helper.LMPCPrediction(lap_number=laps)
    sim = offboard.CarRacingSim(); sim.set_timestep(timestep); sim.set_track(track); sim.add_vehicle(ego)
    sim.set_opti_traj(opti_xglob)
    cars = []
    for index in range(2):
        c = offboard.NoDynamicsModel(name="car%d" % (index + 1), param=base.CarParam(edgecolor="orange"))
        c.set_track(track)
        cars.append(c)
    for c in (pid, mpc_lti, lmpc_ctrl):
        c.set_racing_sim(sim)
    lmpc_ctrl.set_vehicles_track()

    # record every control.lmpc call: inputs, outputs, certificate of the recorded QP
    calls = []
    ref_lmpc = control.lmpc

    def lmpc_rec(xcurv, par, Atv, Btv, Ctv, ss_curv, Qfun, it, lap_length, lap_width, u_old, system_param):
        n0 = len(RECORDS)
        out = ref_lmpc(xcurv, par, Atv, Btv, Ctv, ss_curv, Qfun, it, lap_length, lap_width, u_old, system_param)
        opti, z, info = RECORDS[-1]
        assert len(RECORDS) == n0 + 1
        if not info["success"] and os.environ.get("CRX_GOLDEN_DEBUG") == "stop":
            print("LMPC solve not certified:", {k: v for k, v in info.items()}, flush=True)
            import pickle
            pickle.dump(dict(x=xcurv, A=Atv, B=Btv, C=Ctv, ss=out[2], qfun=out[3], u_old=u_old, z=z), open("/tmp/lmpc_fail.pkl", "wb"))
            raise SystemExit(1)
        calls.append(dict(
            x=np.array(xcurv, float), A=np.array(Atv, float), B=np.array(Btv, float),
            C=np.array(Ctv, float).reshape(len(Ctv), 6), ss=np.array(out[2], float), qfun=np.array(out[3], float),
            u_old=np.array(u_old, float).reshape(2), U=np.array(out[0], float), X=np.array(out[1], float),
            success=info["success"], cert=cert_fields(info), lap=int(it), lap_width=float(lap_width),
            z_tail=np.array(z[6 * 13 + 2 * 12:], float)))
        return out

    control.lmpc = lmpc_rec
    ELASTIC_ON_INFEASIBLE[0] = 1e5
    del RECORDS[:]
    # laps 0 (PID) and 1 (mpc-lti): every solve certified -> full closed-loop fixture
    sim.sim(sim_time=90, one_lap=True, one_lap_name="ego")
    ego.set_ctrl_policy(mpc_lti)
    sim.sim(sim_time=90, one_lap=True, one_lap_name="ego")
    n_lti = len(RECORDS)
    lti_ok = np.array([r[2]["success"] for r in RECORDS])
    lti_u = np.array([r[1][78:80] for r in RECORDS])
    print("laps 0/1: %d + %d steps, mpc-lti solves %d (certified %d)" % (
        len(ego.xcurvs[0]) - 1, len(ego.xcurvs[1]) - 1, n_lti, lti_ok.sum()), flush=True)
    # lap 2 (LMPC).  The reference pins its terminal slack to zero (control.py:689-690), so its QP is
    # infeasible whenever the regression model cannot reach the safe set; what the reference applies
    # then is IPOPT's restoration-phase state, which cannot be restated.  The loop is continued with
    # the elastic stand-in only to harvest further reference-built problem instances; nothing after
    # the first uncertified step is a closed-loop parity target.
    lmpc_ctrl.add_trajectory(ego, 0)
    lmpc_ctrl.add_trajectory(ego, 1)
    ss_after_two = dict(time_ss=np.array(lmpc_ctrl.time_ss), Qfun0=lmpc_ctrl.Qfun[:600].copy(),
                        ss0=lmpc_ctrl.ss_xcurv[:600].copy(), u0=lmpc_ctrl.u_ss[:600].copy())
    ego.set_ctrl_policy(lmpc_ctrl)
    crashed = ""
    try:
        sim.sim(sim_time=float(os.environ.get("CRX_GOLDEN_LMPC_TIME", "12.0")), one_lap=True, one_lap_name="ego")
    except Exception as e:  # the reference's own regression raises once the state leaves its data
        crashed = repr(e)
    control.lmpc = ref_lmpc
    ok = np.array([c["success"] for c in calls])
    first_bad = int(np.argmin(ok)) if not ok.all() else len(ok)
    out = dict(lap_length=track.lap_length, timestep=timestep, crashed=crashed)
    for it in range(2):
        out["lap%d/xcurv" % it] = np.array(ego.xcurvs[it], float)
        out["lap%d/xglob" % it] = np.array(ego.xglobs[it], float)
        out["lap%d/u" % it] = np.array(ego.inputs[it], float)
        out["lap%d/times" % it] = np.array(ego.times[it], float)
    out["lti_success"], out["lti_u"] = lti_ok, lti_u
    for k, v in ss_after_two.items():
        out["ss/" + k] = v
    out["lmpc_success"] = ok
    out["lmpc_first_uncertified"] = first_bad
    for k in ("x", "A", "B", "C", "ss", "qfun", "u_old", "U", "X", "cert", "lap_width", "lap"):
        out["lmpc/" + k] = np.array([c[k] for c in calls])
    np.savez_compressed(os.path.join(OUT, "racing_game.npz"), **out)
    print("racing game: lmpc calls %d, certified %d, first uncertified step %d, stopped by: %s" % (
        len(calls), int(ok.sum()), first_bad, crashed or "time limit"))


if __name__ == "__main__":
    which = sys.argv[1:] or ["mpccbf", "planner", "harness"]
    if "racing_game" in which:
        gen_racing_game()
    if "path" in which:
        gen_path()
    if "closed_loop" in which:
        gen_closed_loop()
    if "harness" in which:
        gen_harness()
    if "mpccbf" in which:
        gen_mpccbf()
    if "planner" in which:
        gen_planner()
