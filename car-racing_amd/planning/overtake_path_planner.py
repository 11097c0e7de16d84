"""OvertakePathPlanner with the reference's class surface (planning/overtake_path_planner.py:14-318):
lateral-offset paths around the vehicles of interest, one small QP per candidate region, all regions of
a step in ONE crx_path_solve call.

The reference constructs it only when `LMPCRacingGame.path_planner` is True, which its __init__ hard-codes
to False (utils/base.py:414); the mirror honours the attribute if a user flips it.
"""
import datetime

import numpy as np

import crx
from crx import abi, hostprep
from planning import planner_helper as ph
from utils.constants import X_DIM


def path_qp_inputs(ego_s, ego_ey, obs_infos, agent_info, cps, bezier_xcurvs, opt_s, opt_ey, N, track_width, lap_length,
                   safety_factor, prediction_factor, veh_length, veh_width):
    """Arrays of crx_path_solve for the V+1 regions of one step (reference solve_optimization_problem,
    :199-297).  obs_infos [V,3] rows (s, max ey, min ey) of the SORTED vehicles."""
    V = obs_infos.shape[0]
    R = V + 1
    front = hostprep.wrap_above(obs_infos[:, 0] + safety_factor * veh_length, lap_length)   # get_agents_range (:184-197)
    rear = hostprep.wrap_above(obs_infos[:, 0] - safety_factor * veh_length, lap_length)
    span = agent_info.max_s + safety_factor * veh_length + prediction_factor * agent_info.max_delta_v - ego_s
    opt = np.zeros((R, N + 1))
    bez = np.zeros((R, N + 1))
    lb = np.full((R, N + 1), -float(track_width))
    ub = np.full((R, N + 1), float(track_width))
    for j in range(N + 1):
        s_tmp = ego_s + span * j / N                                                        # :232-243
        while s_tmp >= lap_length:
            s_tmp = s_tmp - lap_length
        if s_tmp <= opt_s[0]:
            s_tmp = opt_s[0]
        s_tmp = np.clip(s_tmp, bezier_xcurvs[0, 0, 0], bezier_xcurvs[0, -1, 0])
        opt[:, j] = hostprep.interp_clipped(opt_s, opt_ey, s_tmp)                           # func_optimal_ey (:248)
        for r in range(R):
            bez[r, j] = hostprep.interp_clipped(bezier_xcurvs[r, :, 0], bezier_xcurvs[r, :, 1], s_tmp)   # :250
            if r > 0 and not (s_tmp < rear[r - 1] or s_tmp > front[r - 1]):                 # left neighbour (:266-281)
                bound = obs_infos[r - 1, 2] - safety_factor * veh_width
                if not (j == 0 and ego_ey >= bound):
                    ub[r, j] = min(ub[r, j], bound)
            if r < V and not (s_tmp < rear[r] or s_tmp > front[r]):                          # right neighbour (:283-297)
                bound = obs_infos[r, 1] + safety_factor * veh_width
                if not (j == 0 and ego_ey <= bound):
                    lb[r, j] = max(lb[r, j], bound)
    return opt, bez, lb, ub, np.full(R, float(ego_ey)), np.array(cps[:, 3, 1], dtype=float)


class OvertakePathPlanner:
    def __init__(self, racing_game_param):
        self.racing_game_param = racing_game_param
        self.vehicles = None
        self.agent_name = None
        self.track = None
        self.opti_traj_xcurv = None

    def get_overtake_flag(self, xcurv_ego):
        interest = {}
        for name in list(self.vehicles):
            if name != self.agent_name and ph.check_ego_agent_distance(
                    self.vehicles[self.agent_name], self.vehicles[name], self.racing_game_param, self.track.lap_length):
                interest[name] = self.vehicles[name]
        return bool(interest), interest

    def get_local_path(self, xcurv_ego, time, vehicles_interest):
        start = datetime.datetime.now()
        p = self.racing_game_param
        N = p.num_horizon_planner
        vehicles, track, opt = self.vehicles, self.track, self.opti_traj_xcurv
        ego = vehicles[self.agent_name]
        names = list(vehicles_interest)
        V = len(names)
        order = ph.sort_by_ey(names, lambda n: vehicles_interest[n].xcurv[5])               # :48-55 (quirk Q3)
        agent_info = ph.get_agent_info(vehicles, order, track)
        obs_infos = np.zeros((V, 3))
        for idx, name in enumerate(order):                                                   # SORTED order here (:58-73)
            if vehicles[name].no_dynamics:
                traj, _ = vehicles[name].get_trajectory_nsteps(time, p.timestep, N + 1)
            else:
                traj, _ = vehicles[name].get_trajectory_nsteps(N + 1)
            obs_infos[idx, :] = (vehicles[name].xcurv[4], max(traj.T[:, 5]), min(traj.T[:, 5]))
        cps = ph.get_bezier_control_points(vehicles_interest, obs_infos, agent_info, p, track, opt, order, xcurv_ego)
        bezier_xcurvs = ph.bezier_polylines(cps, N)
        L_ego, W_ego = ego.param.length, ego.param.width
        qp = path_qp_inputs(ego.xcurv[4], ego.xcurv[5], obs_infos, agent_info, cps, bezier_xcurvs, opt[:, 4], opt[:, 5], N,
                            track.width, track.lap_length, p.safety_factor, p.planning_prediction_factor, L_ego, W_ego)
        r = crx.path_solve(abi.path_desc(N, p.alpha), *qp)
        costs = [float(c) for c in r["cost"]]                                                # inf where the QP failed (:311)
        direction_flag = costs.index(min(costs))                                             # first minimum (:313)
        if min(costs) == float("inf"):
            print("path planner failed")
        best_ey = r["E"][direction_flag]
        # target trajectory: s samples, chosen offsets, constant-acceleration speed profile (:115-143)
        span = (agent_info.max_s + p.safety_factor * vehicles["ego"].param.length
                + p.planning_prediction_factor * agent_info.max_delta_v - ego.xcurv[4])
        target = np.zeros((N + 1, X_DIM))
        for j in range(N + 1):
            target[j, 4] = ego.xcurv[4] + span * j / N
            target[j, 5] = best_ey[j]
        s_end = target[-1, 4]
        if s_end >= track.lap_length:
            s_q = opt[0, 4] if s_end - track.lap_length < opt[0, 4] else s_end - track.lap_length
        elif s_end < opt[0, 4]:
            s_q = opt[0, 4]
        else:
            s_q = s_end
        if s_q > opt[-1, 4]:
            raise ValueError("A value in x_new is above the interpolation range.")          # interp1d bounds_error
        vx_target = hostprep.interp_clipped(opt[:, 4], opt[:, 0], s_q)
        delta_t = 2 * (s_end - xcurv_ego[4]) / (vx_target + xcurv_ego[0])
        a_target = np.clip((vx_target - xcurv_ego[0]) / delta_t, -1.5, 1.5)
        target = self.get_speed_info(target, xcurv_ego, a_target)
        solver_time = (datetime.datetime.now() - start).total_seconds()
        print("local planner solver time: {}".format(solver_time))
        target_xglob = ph.get_traj_xglob(target, track)
        line = np.zeros((N + 1, X_DIM))
        line[:, 4:6] = bezier_xcurvs[direction_flag]
        bezier_xglob = ph.get_traj_xglob(line, track)
        all_bezier_xglob = np.zeros((V + 1, N + 1, X_DIM))
        for idx in range(V + 1):
            line = np.zeros((N + 1, X_DIM))
            line[:, 4:6] = bezier_xcurvs[idx]
            all_bezier_xglob[idx] = ph.get_traj_xglob(line, track)
        all_local_traj_xglob = np.zeros((V + 1, N + 1, X_DIM))                               # never filled by the reference (:160)
        return (target, target_xglob, direction_flag, order, bezier_xglob, solver_time, all_bezier_xglob,
                all_local_traj_xglob)

    def get_speed_info(self, target_traj_xcurv, xcurv, a_target):
        """First row = the current state, speeds from v^2 = v0^2 + 2 a (s - s0) for rows 0..N-1 (:185-194)."""
        N = self.racing_game_param.num_horizon_planner
        traj = target_traj_xcurv
        traj[0, :] = xcurv
        for j in range(N):
            traj[j, 0] = (xcurv[0] ** 2 + 2 * a_target * (traj[j, 4] - xcurv[4])) ** 0.5
        return traj

    def get_agents_range(self, num_veh, obs_traj_infos):
        """(:196-211)."""
        p, L = self.racing_game_param, self.vehicles[self.agent_name].param.length
        front = hostprep.wrap_above(obs_traj_infos[:num_veh, 0] + p.safety_factor * L, self.track.lap_length)
        rear = hostprep.wrap_above(obs_traj_infos[:num_veh, 0] - p.safety_factor * L, self.track.lap_length)
        return front.reshape(num_veh, 1), rear.reshape(num_veh, 1)
