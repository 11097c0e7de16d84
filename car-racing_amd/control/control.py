"""Controller front-ends with the reference's call signatures (control/control.py), whose solver
bodies marshal arrays into the crx C ABI instead of building a CasADi Opti problem.

  mpccbf            <- reference control/control.py:476-607
  mpc_multi_agents  <- reference control/control.py:251-473
  mpc_lti           <- reference control/control.py:198-248   (same NLP family with no obstacle)
  lmpc              <- reference control/control.py:610-730   (QP on the GPU, crx_lmpc_solve)
  pid               <- reference control/control.py:15-25

There is no CPU path: without libcrx / a GPU these raise crx.CrxUnavailable.
"""
import datetime

import numpy as np

import crx
from crx import abi, hostprep
from utils.constants import U_DIM, X_DIM

_N_OBS_MAX = abi.CRX_MAX_OBS


def pid(xcurv, xtarget):
    """P steering on (ey, epsi), P acceleration on vx (reference :15-25)."""
    xt = np.asarray(xtarget, dtype=float).reshape(-1)
    u = np.zeros((U_DIM,))
    u[0] = -0.6 * (xcurv[5] - xt[5]) - 0.9 * xcurv[3]
    u[1] = 1.5 * (xt[0] - xcurv[0])
    return u


def _predictions(vehicles, names, time, timestep, n, realtime_flag):
    """obs_traj of every named vehicle (reference :506-516 / :295-303)."""
    out = []
    for name in names:
        if realtime_flag is False:
            traj, _ = vehicles[name].get_trajectory_nsteps(time, timestep, n)
        else:
            traj, _ = vehicles[name].get_trajectory_nsteps(n)
        out.append(np.asarray(traj, dtype=float))
    return out


def _solve_cbf(desc, x, xt, preds, lap_length, dims=None):
    """Window filter (:499-523), lap offsets (:538-540, quirk Q1) and one batched crx call of size 1.  dims: (l_sum, w_sum) per
    entry of `preds` when the obstacle vehicles differ in size (:529-535), None = the descriptor's pair."""
    N = desc.N
    V = len(preds)
    obs_s = np.zeros((1, max(V, 1), N + 1))
    obs_ey = np.zeros((1, max(V, 1), N + 1))
    for v, tr in enumerate(preds):
        obs_s[0, v], obs_ey[0, v] = tr[4, :], tr[5, :]
    x = np.asarray(x, dtype=float).reshape(1, X_DIM)
    if V:
        keep, lap_off = hostprep.cbf_window(x, obs_s[:, :V, 0], lap_length)
    else:
        keep, lap_off = np.zeros((1, 0), dtype=bool), np.zeros((1, 0))
    nmax = desc.n_obs_max
    if nmax and dims is not None:
        ps, pe, po, n, pd = hostprep.pack_obstacles(keep, obs_s[:, :V], obs_ey[:, :V], lap_off, nmax, ego_s=x[:, 4],
                                                    dims=np.asarray(dims, dtype=float).reshape(1, V, 2))
        return crx.cbf_solve(desc, x, xt, ps, pe, po, n, obs_dims=pd)
    ps, pe, po, n = hostprep.pack_obstacles(keep, obs_s[:, :V], obs_ey[:, :V], lap_off, nmax, ego_s=x[:, 4]) if nmax else (
        np.zeros((1, 0, N + 1)), np.zeros((1, 0, N + 1)), np.zeros((1, 0)), np.zeros(1, dtype=np.int32))
    return crx.cbf_solve(desc, x, xt, ps, pe, po, n)


def mpc_lti(xcurv, xtarget, mpc_lti_param, system_param, track):
    N = mpc_lti_param.num_horizon
    desc = abi.cbf_desc(
        N, 0, mpc_lti_param.matrix_A, mpc_lti_param.matrix_B, Q=np.diag(mpc_lti_param.matrix_Q),
        R=np.diag(mpc_lti_param.matrix_R), ey_max=track.width, delta_max=system_param.delta_max,
        a_max=system_param.a_max, v_min=system_param.v_min, v_max=system_param.v_max)
    xt = np.asarray(xtarget, dtype=float).reshape(1, X_DIM)
    r = _solve_cbf(desc, xcurv, xt, [], track.lap_length)
    if r["status"][0] != abi.CRX_CONVERGED:
        # the reference has no handler here: an IPOPT failure propagates (:242)
        raise RuntimeError("mpc_lti: solver status %d" % int(r["status"][0]))
    return r["U"][0, 0, :]


def mpccbf(xcurv, xtarget, mpc_cbf_param, vehicles, agent_name, lap_length, time, timestep, realtime_flag,
           track, system_param):
    start = datetime.datetime.now()
    N = mpc_cbf_param.num_horizon
    others = [n for n in list(vehicles) if n != agent_name]
    preds = _predictions(vehicles, others, time, timestep, N + 1, realtime_flag)
    if len(others) > 3:
        # more vehicles than the tuned instantiations carry (3): only those that pass the window test enter the NLP anyway
        # (:499-523), so filter first -- up to CRX_MAX_OBS = 6 of them are solved exactly (generic instantiation beyond 3),
        # more than that and pack_obstacles keeps the nearest and warns
        x1 = np.asarray(xcurv, dtype=float).reshape(1, X_DIM)
        keep, _ = hostprep.cbf_window(x1, np.array([[p[4, 0] for p in preds]]), lap_length)
        preds = [p for p, k in zip(preds, keep[0]) if k]
        others = [n for n, k in zip(others, keep[0]) if k]
    ego, first = vehicles[agent_name], (vehicles[others[0]] if others else vehicles[agent_name])
    # (l, w) per obstacle (control.py:529-535): crx_cbf_solve_dims when the cars differ, the descriptor's pair when they do not
    same = all((vehicles[n].param.length, vehicles[n].param.width) == (first.param.length, first.param.width) for n in others)
    dims = None if same else [(ego.param.length / 2 + vehicles[n].param.length / 2, ego.param.width / 2 + vehicles[n].param.width / 2)
                              for n in others]
    desc = abi.cbf_desc(
        N, min(len(preds), _N_OBS_MAX), mpc_cbf_param.matrix_A, mpc_cbf_param.matrix_B,
        Q=np.diag(mpc_cbf_param.matrix_Q), R=np.diag(mpc_cbf_param.matrix_R), alpha=mpc_cbf_param.alpha,
        margin=0.2, ey_max=track.width, delta_max=system_param.delta_max, a_max=system_param.a_max,
        v_min=system_param.v_min, v_max=system_param.v_max,
        l_sum=ego.param.length / 2 + first.param.length / 2, w_sum=ego.param.width / 2 + first.param.width / 2)
    xt = np.asarray(xtarget, dtype=float).reshape(1, X_DIM)
    r = _solve_cbf(desc, xcurv, xt, preds, lap_length, dims=dims)
    if r["status"][0] != abi.CRX_CONVERGED:
        print("solver failed.")  # the reference then uses the last iterate (:600-603); so do we
    print("solver time: {}".format((datetime.datetime.now() - start).total_seconds()))
    return r["U"][0, 0, :]


def mpc_multi_agents(xcurv, mpc_lti_param, track, matrix_Atv, matrix_Btv, matrix_Ctv, system_param,
                     target_traj_xcurv=None, vehicles=None, agent_name=None, direction_flag=None,
                     target_traj_xglob=None, sorted_vehicles=None, time=None):
    print("overtaking")
    start = datetime.datetime.now()
    N = mpc_lti_param.num_horizon_ctrl
    names = [n for n in sorted_vehicles if n != agent_name]
    preds = _predictions(vehicles, names, time, 0.1, N + 1, False)  # literals of the reference (:288-290)
    ego = vehicles["ego"]
    desc = abi.cbf_desc(
        N, min(len(preds), _N_OBS_MAX), mpc_lti_param.matrix_A, mpc_lti_param.matrix_B,
        Q=np.diag(mpc_lti_param.matrix_Q), R=np.diag(mpc_lti_param.matrix_R), alpha=0.6, margin=0.15,
        ey_max=track.width, per_stage_target=True, delta_max=system_param.delta_max, a_max=system_param.a_max,
        v_min=system_param.v_min, v_max=system_param.v_max, l_sum=ego.param.length, w_sum=ego.param.width)
    xt = hostprep.tracking_targets(xcurv, np.asarray(target_traj_xcurv, dtype=float), N)[None]
    r = _solve_cbf(desc, xcurv, xt, preds, track.lap_length)
    if r["status"][0] != abi.CRX_CONVERGED:
        print("solver fail")
    print("solver time: {}".format((datetime.datetime.now() - start).total_seconds()))
    return r["U"][0, 0, :], r["X"][0]


def _out_of_scope(name):
    def f(*a, **k):
        raise NotImplementedError(
            "%s is outside the accelerated hot path (SURVEY.md section 8f, 'next' rows); use the reference" % name)
    f.__name__ = name
    return f


def lmpc(xcurv, lmpc_param, matrix_Atv, matrix_Btv, matrix_Ctv, ss_curv, Qfun, iter, lap_length, lap_width, u_old,
         system_param):
    """Learning MPC step (reference control/control.py:610-730): safe-set selection on the host
    (:625-639), the QP on the GPU (crx_lmpc_solve).  Returns (u_pred (N,2), x_pred (N+1,6),
    ss_point_selected_tot (6,M), Qfun_selected_tot (M,), lin_points (N+1,6), lin_input (N,2)) like
    the reference.  Where the reference's QP is infeasible (its terminal slack is pinned to zero,
    :694-695) the reference applies IPOPT's restoration state; here the plan from a minimally
    relaxed initial state comes back instead (include/crx.h, crx_lmpc_solve) and the same message is
    printed."""
    from control import lmpc_helper

    start = datetime.datetime.now()
    N = lmpc_param.num_horizon
    pts, qs = [], []
    for jj in range(lmpc_param.num_ss_iter):
        p, q = lmpc_helper.select_points(ss_curv, Qfun, iter - jj - 1, xcurv,
                                         lmpc_param.num_ss_points / lmpc_param.num_ss_iter, lmpc_param.shift)
        pts.append(p)
        qs.append(q)
    ss_sel, q_sel = np.concatenate(pts, axis=1), np.concatenate(qs, axis=0)
    M = q_sel.shape[0]
    desc = abi.lmpc_desc(
        N=N, n_ss_max=M, Q=np.diag(lmpc_param.matrix_Q), R=np.diag(lmpc_param.matrix_R),
        dR=np.diag(lmpc_param.matrix_dR), v_max=system_param.v_max, ey_max=lap_width,
        delta_max=system_param.delta_max, a_max=system_param.a_max)
    r = crx.lmpc_solve(desc, np.asarray(xcurv, dtype=float)[None], np.asarray(u_old, dtype=float).reshape(1, 2),
                       np.asarray(matrix_Atv, dtype=float)[None], np.asarray(matrix_Btv, dtype=float)[None],
                       np.asarray(matrix_Ctv, dtype=float).reshape(1, N, X_DIM), ss_sel[None], q_sel[None])
    if r["status"][0] != 0:
        print("solver fail to find the solution, the non-converged solution is used")
    x_pred, u_pred = r["X"][0], r["U"][0]
    lin_points = np.concatenate((x_pred[1:, :], x_pred[-1:, :]), axis=0)
    lin_input = np.vstack((u_pred[1:, :], u_pred[-1, :]))
    print("solver time: {}".format((datetime.datetime.now() - start).total_seconds()))
    return u_pred, x_pred, ss_sel, q_sel, lin_points, lin_input


lqr = _out_of_scope("lqr")
ilqr = _out_of_scope("ilqr")
