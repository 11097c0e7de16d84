"""Controller / vehicle / simulator classes with the reference's surface (utils/base.py), so that
the reference's drivers (tests/auto_mpccbf_test.py, car_racing/tests/mpccbf_test.py) run against
this package unchanged.  LQR and iLQR are out of scope and raise.

Objects stay picklable: the GPU library handle lives in the `crx` module, never in instance state
(the reference pickles its simulator, tests/auto_mpccbf_test.py:42-43).
"""
import copy
import os

import numpy as np

from control import control, lmpc_helper
from planning import overtake_path_planner, overtake_traj_planner
from system import vehicle_dynamics
from utils import racing_env
from utils.constants import U_DIM, X_DIM

_REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def _csv(rel):
    """The reference reads its model CSVs relative to the CWD at import time (base.py:124-125);
    fall back to this repository's copy of the same data files."""
    path = rel if os.path.exists(rel) else os.path.join(_REPO, rel)
    return np.genfromtxt(path, delimiter=",")


_A_DEFAULT = _csv("data/sys/LTI/matrix_A.csv")
_B_DEFAULT = _csv("data/sys/LTI/matrix_B.csv")


class ControlBase:
    def __init__(self):
        self.agent_name = None
        self.time = 0.0
        self.timestep = None
        self.x = None
        self.xglob = None
        self.u = None
        self.realtime_flag = False
        self.lap_times, self.lap_xcurvs, self.lap_xglobs, self.lap_inputs = [self.time], [], [], []
        self.times, self.xglobs, self.xcurvs, self.inputs = [], [], [], []
        self.laps = 0
        self.track = None
        self.opti_traj_xcurv = None
        self.opti_traj_xglob = None

    def set_track(self, track):
        self.track = track
        self.lap_length = track.lap_length
        self.point_and_tangent = track.point_and_tangent
        self.lap_width = track.width

    def set_opti_traj(self, opti_traj_xcurv, opti_traj_xglob):
        self.opti_traj_xcurv, self.opti_traj_xglob = opti_traj_xcurv, opti_traj_xglob

    def set_racing_sim(self, racing_sim):
        self.racing_sim = racing_sim

    def set_timestep(self, timestep):
        self.timestep = timestep

    def set_target_speed(self, vt):
        self.vt = vt

    def set_target_deviation(self, eyt):
        self.eyt = eyt

    def set_state(self, xcurv, xglob):
        self.x, self.xglob = xcurv, xglob

    def calc_input(self):
        pass

    def get_input(self):
        return self.u

    def _ego_log_none(self, **kw):
        """Per-step bookkeeping the reference appends to vehicles['ego'] (base.py:106-117,336-347)."""
        if self.agent_name != "ego":
            return
        veh = (self.racing_sim.vehicles if self.realtime_flag is False else self.vehicles)["ego"]
        for key in ("local_trajs", "vehicles_interest", "splines", "all_splines", "all_local_trajs",
                    "lmpc_prediction", "mpc_cbf_prediction"):
            getattr(veh, key).append(kw.get(key))


class PIDTracking(ControlBase):
    def __init__(self, vt=0.6, eyt=0.0):
        ControlBase.__init__(self)
        self.set_target_speed(vt)
        self.set_target_deviation(eyt)

    def calc_input(self):
        xtarget = np.array([self.vt, 0, 0, 0, 0, self.eyt]).reshape(X_DIM, 1)
        self.u = control.pid(self.x, xtarget)
        self._ego_log_none()
        self.time += self.timestep


class MPCTrackingParam:
    def __init__(self, matrix_A=_A_DEFAULT, matrix_B=_B_DEFAULT, matrix_Q=np.diag([10.0, 0.0, 0.0, 4.0, 0.0, 40.0]),
                 matrix_R=np.diag([0.1, 0.1]), vt=0.6, eyt=0.0, num_horizon=10):
        self.matrix_A, self.matrix_B, self.matrix_Q, self.matrix_R = matrix_A, matrix_B, matrix_Q, matrix_R
        self.vt, self.eyt, self.num_horizon = vt, eyt, num_horizon


class MPCTracking(ControlBase):
    def __init__(self, mpc_lti_param, system_param):
        ControlBase.__init__(self)
        self.set_target_speed(mpc_lti_param.vt)
        self.set_target_deviation(mpc_lti_param.eyt)
        self.mpc_lti_param, self.system_param = mpc_lti_param, system_param

    def calc_input(self):
        xtarget = np.array([self.vt, 0, 0, 0, 0, self.eyt]).reshape(X_DIM, 1)
        self.u = control.mpc_lti(self.x, xtarget, self.mpc_lti_param, self.system_param, self.track)
        self._ego_log_none()
        self.time += self.timestep


class MPCCBFRacingParam:
    def __init__(self, matrix_A=_A_DEFAULT, matrix_B=_B_DEFAULT, matrix_Q=np.diag([10.0, 0.0, 0.0, 4.0, 0.0, 40.0]),
                 matrix_R=np.diag([0.1, 0.1]), vt=0.6, eyt=0.0, num_horizon=10, alpha=0.8):
        self.matrix_A, self.matrix_B, self.matrix_Q, self.matrix_R = matrix_A, matrix_B, matrix_Q, matrix_R
        self.vt, self.eyt, self.num_horizon, self.alpha = vt, eyt, num_horizon, alpha


class MPCCBFRacing(ControlBase):
    def __init__(self, mpc_cbf_param, system_param):
        ControlBase.__init__(self)
        self.set_target_speed(mpc_cbf_param.vt)
        self.set_target_deviation(mpc_cbf_param.eyt)
        self.realtime_flag = None
        self.mpc_cbf_param, self.system_param = mpc_cbf_param, system_param

    def calc_input(self):
        xtarget = np.array([self.vt, 0, 0, 0, 0, self.eyt]).reshape(X_DIM, 1)
        if self.realtime_flag is False:
            vehicles, lap_length = self.racing_sim.vehicles, self.racing_sim.track.lap_length
        elif self.realtime_flag is True:
            vehicles, lap_length = self.vehicles, self.lap_length
        else:
            vehicles = None
        if vehicles is not None:
            self.u = control.mpccbf(self.x, xtarget, self.mpc_cbf_param, vehicles, self.agent_name, lap_length,
                                    self.time, self.timestep, self.realtime_flag, self.track, self.system_param)
        self._ego_log_none()
        self.time += self.timestep


class RacingGameParam:
    def __init__(self, matrix_A=_A_DEFAULT, matrix_B=_B_DEFAULT, matrix_Q=np.diag([10.0, 0.0, 0.0, 5.0, 0.0, 50.0]),
                 matrix_R=np.diag([0.1, 0.1]), matrix_R_planner=1 * np.diag([5, 0.10]),
                 matrix_dR_planner=5 * np.diag([1.8, 0.0]), bezier_order=3, safety_factor=4.5, num_horizon_ctrl=10,
                 num_horizon_planner=10, planning_prediction_factor=0.5, alpha=0.98, timestep=None):
        self.matrix_A, self.matrix_B, self.matrix_Q, self.matrix_R = matrix_A, matrix_B, matrix_Q, matrix_R
        self.matrix_R_planner, self.matrix_dR_planner = matrix_R_planner, matrix_dR_planner
        self.num_horizon_ctrl, self.num_horizon_planner = num_horizon_ctrl, num_horizon_planner
        self.planning_prediction_factor, self.alpha, self.timestep = planning_prediction_factor, alpha, timestep
        self.bezier_order, self.safety_factor = bezier_order, safety_factor


class LMPCRacingParam:
    def __init__(self, matrix_Q=0 * np.diag([0.0] * 6), matrix_R=1 * np.diag([1.0, 0.25]),
                 matrix_Qslack=5 * np.diag([10, 0, 0, 1, 10, 0]), matrix_dR=5 * np.diag([0.8, 0.0]),
                 num_ss_points=32 + 12, num_ss_iter=2, num_horizon=12, shift=0, timestep=None, lap_number=None,
                 time_lmpc=None):
        self.matrix_Q, self.matrix_R, self.matrix_Qslack, self.matrix_dR = matrix_Q, matrix_R, matrix_Qslack, matrix_dR
        self.num_ss_points, self.num_ss_iter, self.num_horizon, self.shift = num_ss_points, num_ss_iter, num_horizon, shift
        self.timestep, self.lap_number, self.time_lmpc = timestep, lap_number, time_lmpc


class LMPCRacingGame(ControlBase):
    """The reference's racing-game controller (utils/base.py:410-682): learning MPC on the safe set
    of earlier laps while the road is free (:468-517), overtaking planner + tracking NLP while another
    vehicle is of interest (:518-582).  All three solves (crx_lmpc_solve; crx_planner_plan;
    crx_cbf_solve) run on the GPU; safe-set bookkeeping and the local model regression stay on the
    host (control/lmpc_helper.py)."""

    def __init__(self, lmpc_param, racing_game_param=None, system_param=None, path_planner=False):
        ControlBase.__init__(self)
        # the reference hard-codes False (:414); the keyword lets a user pick its OvertakePathPlanner (:415-416)
        self.path_planner = path_planner
        self.lmpc_param, self.racing_game_param, self.system_param = lmpc_param, racing_game_param, system_param
        if self.path_planner:
            self.overtake_planner = overtake_path_planner.OvertakePathPlanner(racing_game_param)
        else:
            self.overtake_planner = overtake_traj_planner.OvertakeTrajPlanner(racing_game_param)
        self.x_pred = self.u_pred = None
        self.lin_points = self.lin_input = None
        self.ss_point_selected_tot = self.Qfun_selected_tot = None
        laps = lmpc_param.lap_number
        num_points = int(lmpc_param.time_lmpc / lmpc_param.timestep) + 1
        # sampled safe set (:432-443): states, inputs, cost-to-go and completion time of every lap
        self.time_ss = 10000 * np.ones(laps).astype(int)
        self.ss_xcurv = 10000 * np.ones((num_points, X_DIM, laps))
        self.u_ss = 10000 * np.ones((num_points, U_DIM, laps))
        self.Qfun = 0 * np.ones((num_points, laps))
        self.ss_glob = 10000 * np.ones((num_points, X_DIM, laps))
        self.iter = 0
        self.time_in_iter = 0
        self.openloop_prediction = None
        self.old_ey = self.old_direction_flag = None

    def set_vehicles_track(self):
        if self.realtime_flag is False:
            vehicles = self.racing_sim.vehicles
            self.overtake_planner.track = self.track
        else:
            vehicles = self.vehicles
        self.overtake_planner.vehicles = vehicles

    def _prediction_xglob(self, x_pred):
        n = x_pred.shape[0]
        pred_glob = np.zeros((n, X_DIM))
        for j in range(n):
            pred_glob[j, 0:3] = x_pred[j, 0:3]
            pred_glob[j, 3] = self.track.get_orientation(x_pred[j, 4], x_pred[j, 5])
            pred_glob[j, 4], pred_glob[j, 5] = self.track.get_global_position(x_pred[j, 4], x_pred[j, 5])
        return pred_glob

    def calc_input(self):
        pl = self.overtake_planner
        pl.agent_name, pl.opti_traj_xcurv = self.agent_name, self.opti_traj_xcurv
        matrix_Atv, matrix_Btv, matrix_Ctv, _ = self.estimate_ABC()
        x = copy.deepcopy(self.x)
        while x[4] > self.lap_length:
            x[4] = x[4] - self.lap_length
        u_old = np.zeros((1, 2)) if self.u_pred is None else copy.deepcopy(self.u_pred[0, :])
        overtake_flag, vehicles_interest = pl.get_overtake_flag(x)
        ego = pl.vehicles["ego"]
        if not overtake_flag:
            (self.u_pred, self.x_pred, self.ss_point_selected_tot, self.Qfun_selected_tot, self.lin_points,
             self.lin_input) = control.lmpc(x, self.lmpc_param, matrix_Atv, matrix_Btv, matrix_Ctv, self.ss_xcurv,
                                            self.Qfun, self.iter, self.lap_length, self.lap_width, u_old,
                                            self.system_param)
            self.u = self.u_pred[0, :]
            self.old_ey = self.old_direction_flag = None
            log = self.openloop_prediction
            if log is not None:
                log.predicted_xcurv[:, :, self.time_in_iter, self.iter] = self.x_pred
                log.predicted_u[:, :, self.time_in_iter, self.iter] = self.u_pred
                log.ss_used[:, :, self.time_in_iter, self.iter] = self.ss_point_selected_tot
                log.Qfun_used[:, self.time_in_iter, self.iter] = self.Qfun_selected_tot
            self.add_point(self.x, self.u, self.time_in_iter)
            self.time_in_iter = self.time_in_iter + 1
            for lst in (ego.local_trajs, ego.vehicles_interest, ego.splines, ego.solver_time, ego.all_splines,
                        ego.all_local_trajs):
                lst.append(None)
            ego.lmpc_prediction.append(self._prediction_xglob(self.x_pred))
            ego.mpc_cbf_prediction.append(None)
        else:
            if self.path_planner:
                (traj_xcurv, traj_xglob, direction_flag, sorted_vehicles, bezier_xglob, solve_time, all_bezier_xglob,
                 all_traj_xglob) = pl.get_local_path(x, self.time, vehicles_interest)
            else:
                (traj_xcurv, traj_xglob, direction_flag, sorted_vehicles, bezier_xglob, solve_time, all_bezier_xglob,
                 all_traj_xglob) = pl.get_local_traj(x, self.time, vehicles_interest, matrix_Atv, matrix_Btv, matrix_Ctv,
                                                     self.old_ey, self.old_direction_flag)
            self.old_ey, self.old_direction_flag = traj_xcurv[-1, 5], direction_flag
            ego.local_trajs.append(traj_xglob)
            ego.vehicles_interest.append(vehicles_interest)
            ego.splines.append(bezier_xglob)
            ego.solver_time.append(solve_time)
            ego.all_splines.append(all_bezier_xglob)
            ego.all_local_trajs.append(all_traj_xglob)
            self.u, x_pred = control.mpc_multi_agents(
                x, self.racing_game_param, self.track, matrix_Atv, matrix_Btv, matrix_Ctv, self.system_param,
                target_traj_xcurv=traj_xcurv, vehicles=pl.vehicles, agent_name=self.agent_name,
                direction_flag=direction_flag, target_traj_xglob=traj_xglob, sorted_vehicles=sorted_vehicles)
            ego.lmpc_prediction.append(None)
            ego.mpc_cbf_prediction.append(self._prediction_xglob(x_pred))
        self.time += self.timestep

    def estimate_ABC(self):
        """One local model per horizon stage from the two previous laps (:585-622)."""
        used_iter = range(self.iter - 2, self.iter)
        Atv, Btv, Ctv, index_used = [], [], [], []
        last = getattr(self, "_last_models", None)
        for i in range(self.lmpc_param.num_horizon):
            try:
                Ai, Bi, Ci, idx = lmpc_helper.regression_and_linearization(
                    self.lin_points, self.lin_input, used_iter, self.ss_xcurv, self.u_ss, self.time_ss, 40, None, None,
                    self.point_and_tangent, self.timestep, i)
            except np.linalg.LinAlgError:
                # no stored sample within the bandwidth of this linearisation point: the reference hands cvxopt a singular
                # normal matrix here and raises.  lmpc_helper.ON_SINGULAR = "keep" keeps the previous model of the stage.
                if lmpc_helper.ON_SINGULAR != "keep" or last is None:
                    raise
                print("local regression singular at stage %d: previous stage model kept" % i)
                Ai, Bi, Ci, idx = last[i]
            Atv.append(Ai)
            Btv.append(Bi)
            Ctv.append(Ci)
            index_used.append(idx)
        self._last_models = list(zip(Atv, Btv, Ctv, index_used))
        return Atv, Btv, Ctv, index_used

    def add_point(self, x, u, i):
        """Extend the previous lap's safe set past the finish line with the running lap (:624-629)."""
        counter = self.time_ss[self.iter - 1]
        self.ss_xcurv[counter + i + 1, :, self.iter - 1] = x + np.array([0, 0, 0, 0, self.lap_length, 0])
        self.u_ss[counter + i + 1, :, self.iter - 1] = u[:]

    def add_trajectory(self, ego, lap_number):
        """Store a completed lap in the safe set (:631-656)."""
        it = self.iter
        end_iter = int(round((ego.times[lap_number][-1] - ego.times[lap_number][0]) / ego.timestep))
        self.time_ss[it] = end_iter
        xcurvs = np.stack(ego.xcurvs[lap_number], axis=0)
        xglobs = np.stack(ego.xglobs[lap_number], axis=0)
        inputs = np.stack(ego.inputs[lap_number], axis=0)
        self.ss_xcurv[0:end_iter + 1, :, it] = xcurvs[0:end_iter + 1, :]
        self.ss_glob[0:end_iter + 1, :, it] = xglobs[0:end_iter + 1, :]
        self.u_ss[0:end_iter, :, it] = inputs[0:end_iter, :]
        self.Qfun[0:end_iter + 1, it] = lmpc_helper.compute_cost(xcurvs[0:end_iter + 1, :], inputs[0:end_iter, :],
                                                                  self.lap_length)
        # beyond the finish line the cost-to-go keeps counting down (:647-649)
        q = self.Qfun[:, it]
        for i in np.nonzero(q[0:end_iter + 1] == 0)[0]:
            q[i] = q[i - 1] - 1
        q[end_iter + 1:] = q[end_iter] - np.arange(1, q.shape[0] - end_iter)
        if self.iter == 0:
            self.lin_points = self.ss_xcurv[1:self.lmpc_param.num_horizon + 2, :, it]
            self.lin_input = self.u_ss[1:self.lmpc_param.num_horizon + 1, :, it]
        self.iter = self.iter + 1
        self.time_in_iter = 0


# ---------------------------------------------------------------------------------------------------
# vehicle models
# ---------------------------------------------------------------------------------------------------
class BicycleDynamicsParam:
    def __init__(self, m=1.98, lf=0.125, lr=0.125, Iz=0.024, Df=0.8 * 1.98 * 9.81 / 2.0, Cf=1.25, Bf=1.0,
                 Dr=0.8 * 1.98 * 9.81 / 2.0, Cr=1.25, Br=1.0):
        self.m, self.lf, self.lr, self.Iz = m, lf, lr, Iz
        self.Df, self.Cf, self.Bf, self.Dr, self.Cr, self.Br = Df, Cf, Bf, Dr, Cr, Br

    def get_params(self):
        return (self.m, self.lf, self.lr, self.Iz, self.Df, self.Cf, self.Bf, self.Dr, self.Cr, self.Br)


class CarParam:
    def __init__(self, length=0.4, width=0.2, facecolor="None", edgecolor="black"):
        self.length, self.width, self.facecolor, self.edgecolor = length, width, facecolor, edgecolor
        self.dynamics_param = BicycleDynamicsParam()


class SystemParam:
    def __init__(self, delta_max=0.5, a_max=1.0, v_max=10, v_min=0):
        self.delta_max, self.a_max, self.v_max, self.v_min = delta_max, a_max, v_max, v_min


class ModelBase:
    def __init__(self, name=None, param=None, no_dynamics=False, system_param=None):
        self.name, self.param, self.system_param = name, param, system_param
        self.no_dynamics = False
        self.time = 0.0
        self.timestep = None
        self.xcurv = self.xglob = self.u = None
        self.zero_noise_flag = False
        self.lap_times = [self.time]
        self.lap_xcurvs, self.lap_xglobs, self.lap_inputs = [], [], []
        self.times, self.xglobs, self.xcurvs, self.inputs = [], [], [], []
        self.laps = 0
        self.realtime_flag = False
        self.xglob_log, self.xcurv_log = [], []
        self.local_trajs, self.vehicles_interest, self.splines, self.solver_time = [], [], [], []
        self.all_splines, self.all_local_trajs, self.lmpc_prediction, self.mpc_cbf_prediction = [], [], [], []

    def set_zero_noise(self):
        self.zero_noise_flag = True

    def set_timestep(self, dt):
        self.timestep = dt

    def set_state_curvilinear(self, xcurv):
        self.xcurv = xcurv

    def set_state_global(self, xglob):
        self.xglob = xglob

    def start_logging(self):
        self.lap_xcurvs, self.lap_xglobs, self.lap_inputs = [self.xcurv], [self.xglob], []

    def set_track(self, track):
        self.track = track
        self.lap_length = track.lap_length
        self.point_and_tangent = track.point_and_tangent
        self.lap_width = track.width

    def set_ctrl_policy(self, ctrl_policy):
        self.ctrl_policy = ctrl_policy
        self.ctrl_policy.agent_name = self.name

    def calc_ctrl_input(self):
        self.ctrl_policy.set_state(self.xcurv, self.xglob)
        self.ctrl_policy.calc_input()
        self.u = self.ctrl_policy.get_input()

    def forward_dynamics(self):
        pass

    def forward_one_step(self, realtime_flag):
        if self.no_dynamics:
            self.forward_dynamics()
            self.update_memory()
        elif realtime_flag is False:
            self.calc_ctrl_input()
            self.forward_dynamics(realtime_flag)
            self.ctrl_policy.set_state(self.xcurv, self.xglob)
            self.update_memory()
        elif realtime_flag is True:
            self.forward_dynamics(realtime_flag)

    def update_memory(self):
        """Lap bookkeeping (reference base.py:795-819): when s passes the lap length the lap's logs are
        archived and s is wrapped IN PLACE."""
        crossed = self.xcurv[4] > self.lap_length
        self.xglob_log.append(self.xglob)
        self.xcurv_log.append(self.xcurv)
        self.lap_xglobs.append(self.xglob)
        self.lap_times.append(self.time)
        self.lap_xcurvs.append(copy.deepcopy(self.xcurv) if crossed else self.xcurv)
        if crossed:
            self.lap_inputs.append(self.u)
            self.xglobs.append(self.lap_xglobs)
            self.times.append(self.lap_times)
            self.xcurvs.append(self.lap_xcurvs)
            self.inputs.append(self.lap_inputs)
            self.xcurv[4] = self.xcurv[4] - self.lap_length
            self.laps += 1
            self.lap_xglobs, self.lap_xcurvs, self.lap_inputs, self.lap_times = [self.xglob], [self.xcurv], [], [self.time]
        else:
            self.lap_inputs.append(self.u)


_MOTION_CACHE = {}


def _compiled_motion(t_symbol, s_func, ey_func):
    """s(t), ey(t) and their derivatives as plain callables.  The reference substitutes into the
    sympy expressions on every call (base.py:860-868); compiling once gives the same numbers.  The
    cache is module-level so that vehicle objects stay picklable."""
    import sympy as sp

    key = (str(t_symbol), sp.srepr(sp.sympify(s_func)), sp.srepr(sp.sympify(ey_func)))
    if key not in _MOTION_CACHE:
        def mk(e):
            f = sp.lambdify(t_symbol, sp.sympify(e), "math")
            return lambda t, f=f: float(f(t))
        _MOTION_CACHE[key] = (mk(s_func), mk(ey_func), mk(sp.diff(s_func, t_symbol)), mk(sp.diff(ey_func, t_symbol)))
    return _MOTION_CACHE[key]


class NoDynamicsModel(ModelBase):
    """Scripted vehicle: s(t), ey(t) given as sympy expressions of `t_symbol` (base.py:847-890)."""

    def __init__(self, name=None, param=None, xcurv=None, xglob=None):
        ModelBase.__init__(self, name=name, param=param)
        self.no_dynamics = True

    def set_state_curvilinear_func(self, t_symbol, s_func, ey_func):
        self.t_symbol, self.s_func, self.ey_func = t_symbol, s_func, ey_func
        self.xcurv, self.xglob = self.get_estimation(0)

    def get_estimation(self, t0):
        fs, fe, fds, fde = _compiled_motion(self.t_symbol, self.s_func, self.ey_func)
        xc = np.zeros((X_DIM,))
        xc[0], xc[1] = fds(t0), fde(t0)
        xc[4], xc[5] = fs(t0), fe(t0)
        xg = np.zeros((X_DIM,))
        xg[0:3] = xc[0:3]
        xg[3] = self.track.get_orientation(xc[4], xc[5])
        xg[4], xg[5] = self.track.get_global_position(xc[4], xc[5])
        return xc, xg

    def get_trajectory_nsteps(self, t0, delta_t, n):
        """n-step prediction from the vehicle's OWN clock; t0 is ignored, as in the reference
        (base.py:879-883, quirk Q6)."""
        xc, xg = np.zeros((X_DIM, n)), np.zeros((X_DIM, n))
        for j in range(n):
            xc[:, j], xg[:, j] = self.get_estimation(self.time + j * delta_t)
        return xc, xg

    def forward_dynamics(self):
        self.time += self.timestep
        self.xcurv, self.xglob = self.get_estimation(self.time)


class DynamicBicycleModel(ModelBase):
    def __init__(self, name=None, param=None, xcurv=None, xglob=None, system_param=None):
        ModelBase.__init__(self, name=name, param=param, system_param=system_param)

    def forward_dynamics(self, realtime_flag):
        """100 explicit Euler sub-steps of 1 ms per 0.1 s control step, then bounded process noise
        on (vx, vy, wz) unless zero-noise (reference base.py:897-942)."""
        delta_t = 0.001
        xg, xc = self.xglob, self.xcurv
        dyn = CarParam().dynamics_param
        i = 0
        while (i + 1) * delta_t <= self.timestep:
            if self.u is not None:
                curv = (self.track.get_curvature(xc[4]) if realtime_flag is False
                        else racing_env.get_curvature(self.lap_length, self.point_and_tangent, xc[4]))
                xg, xc = vehicle_dynamics.vehicle_dynamics(dyn, curv, xg, xc, delta_t, self.u)
            i += 1
        noise = (np.clip(np.random.randn() * 0.01, -0.05, 0.05), np.clip(np.random.randn() * 0.01, -0.1, 0.1),
                 np.clip(np.random.randn() * 0.005, -0.05, 0.05))
        if not (((realtime_flag is True) and (self.u is None)) or self.zero_noise_flag):
            for c in range(3):
                xc[c] = xc[c] + 0.5 * noise[c]
        self.xcurv, self.xglob = xc, xg
        self.time += self.timestep


class CarRacingSim:
    def __init__(self):
        self.track = None
        self.vehicles = {}
        self.opti_traj_xglob = None

    def set_timestep(self, dt):
        self.timestep = dt

    def set_track(self, track):
        self.track = track

    def set_opti_traj(self, opti_traj_xglob):
        self.opti_traj_xglob = opti_traj_xglob
