"""Safe-set bookkeeping and local model regression of the learning MPC (reference
control/lmpc_helper.py), written for numpy instead of cvxopt:

  compute_cost                  <- reference :11-23   cost-to-go (time steps to the finish line)
  select_points                 <- reference :278-293 safe-set points ahead of the nearest neighbour
  regression_and_linearization  <- reference :26-189  one LTV stage model (A_i, B_i, C_i)
  LMPCPrediction, closedloop_data <- reference :296-355

The reference identifies the three velocity rows by a kernel-weighted least squares that it hands to
cvxopt's `qp(Q, b)` WITHOUT constraints (:358-366), i.e. the solution of Q x = -b; here that 5x5 /
4x4 system is solved directly.  The three kinematic rows are the analytic Jacobian of the Euler
step, including the reference's `den * 2` in d s/d ey (:163).  Host prep of the 'next' row
(SURVEY.md section 8f #1); the QP itself goes to libcrx (control.lmpc).
"""
import numpy as np

from utils import racing_env
from utils.constants import U_DIM, X_DIM

# What LMPCRacingGame.estimate_ABC does when a stage's normal matrix is singular (no stored sample within the bandwidth
# of the linearisation point -- the previous plan left the data): "raise" = the reference's behaviour (cvxopt raises on
# the singular system, lmpc_helper.py:358-366); "keep" = keep the stage's previous model and go on.
ON_SINGULAR = "raise"

# reference :42-49,57: bandwidth, features and feature scaling of the local regression
_BANDWIDTH = 5.0
_FEATURES = (0, 1, 2)
_SCALE = np.array([0.1, 1.0, 1.0, 1.0, 1.0])


def compute_cost(xcurv, u, lap_length):
    """Cost-to-go of a stored lap: 0 at the last sample and at every sample past the finish line,
    otherwise one more than its successor (reference :11-23)."""
    n = xcurv.shape[0]
    cost = np.zeros(n)
    for i in range(n - 2, -1, -1):
        cost[i] = cost[i + 1] + 1 if xcurv[i, 4] < lap_length else 0.0
    return cost


def select_points(ss_xcurv, Qfun, iter, x0, num_ss_points, shift):
    """The `num_ss_points` stored samples of lap `iter` that start `shift` after the sample nearest
    to x0 in the 1-norm, with their cost-to-go (reference :278-293)."""
    lap = ss_xcurv[:, :, iter]
    nearest = int(np.argmin(np.abs(lap - np.asarray(x0, dtype=float)[None, :]).sum(axis=1)))
    lo = int(nearest + shift) if nearest + shift >= 0 else nearest
    hi = int(lo + num_ss_points)
    return lap[lo:hi, :].T, Qfun[lo:hi, iter]


def _neighbours(ss_xcurv, u_ss, time_ss, lap, x_lin, max_num_point):
    """Indices and Epanechnikov weights of the stored samples of one lap within the bandwidth
    (reference compute_index :192-226)."""
    n = int(time_ss[lap]) - 1
    data = np.hstack((ss_xcurv[0:n, _FEATURES, lap], u_ss[0:n, :, lap]))
    dist = np.abs((data - x_lin[None, :]) * _SCALE[None, :]).sum(axis=1)
    inside = np.nonzero(dist < _BANDWIDTH)[0]
    idx = np.argsort(dist)[0:max_num_point] if inside.shape[0] >= max_num_point else inside
    return idx, (1.0 - (dist[idx] / _BANDWIDTH) ** 2) * 3.0 / 4.0


def _weighted_fit(ss_xcurv, u_ss, used_iter, picks, input_feature, targets):
    """Rows [A(3) B(1) C] of x+ ~ A x[0:3] + B u[input_feature] + C for each target component, from
    the normal equations M'KM w = M'K y (reference compute_Q_M :229-275, compute_b :338-355,
    lmpc_loc_lin_reg :358-366 with lamb = 0)."""
    rows, weights, nxt = [], [], []
    for lap, (idx, k) in zip(used_iter, picks):
        rows.append(np.hstack((ss_xcurv[idx][:, _FEATURES, lap], u_ss[idx][:, [input_feature], lap],
                               np.ones((idx.shape[0], 1)))))
        weights.append(k)
        nxt.append(ss_xcurv[idx + 1][:, :, lap])
    M, K, Y = np.vstack(rows), np.concatenate(weights), np.vstack(nxt)
    Q = M.T @ (K[:, None] * M)
    return [np.linalg.solve(Q, M.T @ (K * Y[:, c])) for c in targets]


def regression_and_linearization(lin_points, lin_input, used_iter, ss_xcurv, u_ss, time_ss, max_num_point, qp, matrix,
                                 point_and_tangent, dt, i):
    """(A_i, B_i, C_i, index_selected) around lin_points[i], lin_input[i] (reference :26-189).  `qp`
    and `matrix` are accepted for signature compatibility and unused."""
    x0 = np.asarray(lin_points[i, :], dtype=float)
    x_lin = np.hstack((x0[list(_FEATURES)], lin_input[i, :]))
    picks = [_neighbours(ss_xcurv, u_ss, time_ss, lap, x_lin, max_num_point) for lap in used_iter]
    Ai, Bi, Ci = np.zeros((X_DIM, X_DIM)), np.zeros((X_DIM, U_DIM)), np.zeros((X_DIM, 1))
    (w_vx,) = _weighted_fit(ss_xcurv, u_ss, used_iter, picks, 1, [0])          # vx driven by a
    w_vy, w_wz = _weighted_fit(ss_xcurv, u_ss, used_iter, picks, 0, [1, 2])    # vy, wz driven by delta
    for row, w, col in ((0, w_vx, 1), (1, w_vy, 0), (2, w_wz, 0)):
        Ai[row, 0:3], Bi[row, col], Ci[row] = w[0:3], w[3], w[4]
    vx, vy, wz, epsi, s, ey = x0
    if s < 0:
        print("s is negative, here the state: \n", lin_points)
    cur = racing_env.get_curvature(point_and_tangent[-1, 3] + point_and_tangent[-1, 4], point_and_tangent, s)
    den = 1.0 - cur * ey
    ce, se = np.cos(epsi), np.sin(epsi)
    along, across = vx * ce - vy * se, vx * se + vy * ce
    # epsi+ = epsi + dt (wz - along / den * cur)
    Ai[3, :] = [-dt * ce / den * cur, dt * se / den * cur, dt, 1.0 + dt * across / den * cur, 0.0,
                -dt * along / den ** 2 * cur * cur]
    Ci[3] = epsi + dt * (wz - along / den * cur) - Ai[3, :] @ x0
    # s+ = s + dt along / den          (d/d ey as the reference writes it: den * 2, :163)
    Ai[4, :] = [dt * ce / den, -dt * se / den, 0.0, -dt * across / den, 1.0, dt * along / (den * 2) * cur]
    Ci[4] = s + dt * along / den - Ai[4, :] @ x0
    # ey+ = ey + dt across
    Ai[5, :] = [dt * se, dt * ce, 0.0, dt * along, 0.0, 1.0]
    Ci[5] = ey + dt * across - Ai[5, :] @ x0
    return Ai, Bi, Ci, [p[0] for p in picks]


class closedloop_data:
    """Closed-loop log buffers (reference :296-329)."""

    def __init__(self, timestep, sim_time, v0):
        self.timestep = timestep
        self.points = int(sim_time / timestep)
        self.u = np.zeros((self.points, 2))
        self.xcurv = np.zeros((self.points + 1, X_DIM))
        self.xglob = np.zeros((self.points + 1, X_DIM))
        self.sim_points = 0.0
        self.xcurv[0, 0] = v0
        self.xglob[0, 0] = v0

    def update_initial_conditions(self, xcurv, xglob):
        self.xcurv[0, :], self.xglob[0, :] = xcurv, xglob
        self.xcurv[1:, :] = 0.0
        self.xglob[1:, :] = 0.0


class LMPCPrediction:
    """Open-loop predictions and safe-set points used at every step (reference :332-355)."""

    def __init__(self, num_horizon=12, points_lmpc=5000, num_ss_points=32 + 12, lap_number=None):
        self.predicted_xcurv = np.zeros((num_horizon + 1, X_DIM, points_lmpc, lap_number))
        self.predicted_u = np.zeros((num_horizon, U_DIM, points_lmpc, lap_number))
        self.ss_used = np.zeros((X_DIM, num_ss_points, points_lmpc, lap_number))
        self.Qfun_used = np.zeros((num_ss_points, points_lmpc, lap_number))
