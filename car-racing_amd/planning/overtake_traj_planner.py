"""OvertakeTrajPlanner with the reference's class surface (planning/overtake_traj_planner.py:11-379).

The reference forks one process per region, each building and solving a CasADi problem, and
exchanges results through a Manager dict (:177-204).  Here every region of the step goes to the GPU
in ONE batched crx_planner_solve call followed by crx_select (a HIP context does not survive fork()).
"""
import datetime

import numpy as np

import crx
from crx import abi, hostprep
from planning import planner_helper as ph
from utils.constants import X_DIM


class OvertakeTrajPlanner:
    def __init__(self, racing_game_param):
        self.racing_game_param = racing_game_param
        self.vehicles = None
        self.agent_name = None
        self.track = None
        self.opti_traj_xcurv = None
        self.matrix_Atv = self.matrix_Btv = self.matrix_Ctv = None
        self.sorted_vehicles = None
        self.obs_infos = None
        self.old_ey = None
        self.old_direction_flag = None
        self.bezier_xcurvs = None
        self.bezier_funcs = None
        self.xcurv_ego = None

    def get_overtake_flag(self, xcurv_ego):
        interest = {}
        for name in list(self.vehicles):
            if name != self.agent_name and ph.check_ego_agent_distance(
                    self.vehicles[self.agent_name], self.vehicles[name], self.racing_game_param, self.track.lap_length):
                interest[name] = self.vehicles[name]
        return bool(interest), interest

    def get_local_traj(self, xcurv_ego, time, vehicles_interest, matrix_Atv, matrix_Btv, matrix_Ctv, old_ey,
                       old_direction_flag):
        self.matrix_Atv, self.matrix_Btv, self.matrix_Ctv = matrix_Atv, matrix_Btv, matrix_Ctv
        start = datetime.datetime.now()
        N = self.racing_game_param.num_horizon_planner
        vehicles, track = self.vehicles, self.track
        names = list(vehicles_interest)
        num_veh = len(names)
        # partial ey-"sort" (:70-76, quirk Q3), predictions and per-vehicle info in ITERATION order (:87-92, Q4)
        order = ph.sort_by_ey(names, lambda n: vehicles_interest[n].xcurv[5])
        obs_infos, veh_infos = {}, np.zeros((num_veh, 3))
        for idx, name in enumerate(names):
            if vehicles[name].no_dynamics:
                traj, _ = vehicles[name].get_trajectory_nsteps(time, self.racing_game_param.timestep, N + 1)
            else:
                traj, _ = vehicles[name].get_trajectory_nsteps(N + 1)
            obs_infos[name] = traj
            veh_infos[idx, :] = (vehicles[name].xcurv[4], max(traj.T[:, 5]), min(traj.T[:, 5]))
        agent_info = ph.get_agent_info(vehicles, order, track)
        cps = ph.get_bezier_control_points(vehicles_interest, veh_infos, agent_info, self.racing_game_param, track,
                                           self.opti_traj_xcurv, order, xcurv_ego)
        self.bezier_xcurvs = ph.bezier_polylines(cps, N)
        self.bezier_funcs = [
            (lambda s, r=r: hostprep.interp_clipped(self.bezier_xcurvs[r, :, 0], self.bezier_xcurvs[r, :, 1], s))
            for r in range(num_veh + 1)]
        self.sorted_vehicles, self.obs_infos = order, obs_infos
        self.old_ey, self.old_direction_flag, self.xcurv_ego = old_ey, old_direction_flag, xcurv_ego
        traj_xcurv, direction_flag, solve_time, solution_xvar = self.solve_optimization_problem()
        print("local planner solver time: {}".format((datetime.datetime.now() - start).total_seconds()))
        traj_xglob = ph.get_traj_xglob(traj_xcurv, track)
        line = np.zeros((N + 1, X_DIM))
        line[:, 4:6] = self.bezier_xcurvs[direction_flag]
        bezier_xglob = ph.get_traj_xglob(line, track)
        all_bezier_xglob = np.zeros((num_veh + 1, N + 1, X_DIM))
        all_traj_xglob = np.zeros((num_veh + 1, N + 1, X_DIM))
        for r in range(num_veh + 1):
            line = np.zeros((N + 1, X_DIM))
            line[:, 4:6] = self.bezier_xcurvs[r]
            all_bezier_xglob[r] = ph.get_traj_xglob(line, track)
            all_traj_xglob[r] = ph.get_traj_xglob(solution_xvar[r].T, track)
        return (traj_xcurv, traj_xglob, direction_flag, order, bezier_xglob, solve_time, all_bezier_xglob,
                all_traj_xglob)

    def solve_optimization_problem(self):
        """All regions of this step in one batch: returns (traj_xcurv (N+1,6), direction_flag,
        solve_time (V+1,), solution_xvar (V+1,6,N+1)) like the reference (:162-246)."""
        p = self.racing_game_param
        N = p.num_horizon_planner
        V = len(self.sorted_vehicles)
        R = V + 1
        ego, track = self.vehicles[self.agent_name], self.track
        obs_s = np.zeros((1, max(V, 1), N + 1))
        obs_ey = np.zeros((1, max(V, 1), N + 1))
        for v, name in enumerate(self.sorted_vehicles):
            obs_s[0, v], obs_ey[0, v] = self.obs_infos[name][4, :], self.obs_infos[name][5, :]
        lb, ub = hostprep.planner_ey_bounds(
            np.asarray(self.xcurv_ego, dtype=float)[None], obs_s[:, :V], obs_ey[:, :V], np.array([V]), track.width,
            track.lap_length, N, veh_length=ego.param.length, veh_width=ego.param.width)
        desc = abi.planner_desc(N, p.matrix_A, p.matrix_B)
        x0 = np.repeat(np.asarray(ego.xcurv, dtype=float)[None], R, axis=0)  # raw state (:266, quirk Q5)
        t0 = datetime.datetime.now()
        # one fused call: every region QP (:248-379) and the selection (:205-246)
        res = crx.planner_plan(
            desc, abi.select_desc(N, V, track.lap_length, ego.param.length, ego.param.width), x0,
            self.bezier_xcurvs[:, :, 0], self.bezier_xcurvs[:, :, 1], lb[0], ub[0], np.array([V]),
            obs_s[:, :V], obs_ey[:, :V],
            np.array([-1 if self.old_direction_flag is None else int(self.old_direction_flag)]))
        sel = res
        dt = (datetime.datetime.now() - t0).total_seconds()
        # the reference wraps the stored predictions in place while selecting (:216-217,:230-231)
        for name in self.sorted_vehicles:
            self.obs_infos[name][4, :] = hostprep.wrap_above(self.obs_infos[name][4, :], track.lap_length)
        direction_flag = int(sel["flag"][0])
        solution_xvar = np.transpose(res["X"], (0, 2, 1)).copy()
        return sel["best_X"][0], direction_flag, np.full(R, dt / R), solution_xvar

    def generate_traj_per_region(self, pos_index, dict_traj, dict_solve_time, dict_cost):
        """Kept for signature compatibility (:248); a single region through the same batched path."""
        traj, _, times, sol = self.solve_optimization_problem()
        dict_traj[pos_index] = sol[pos_index]
        dict_solve_time[pos_index] = times[pos_index]
        dict_cost[pos_index] = None
