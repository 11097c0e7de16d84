"""Host-side helpers of the overtake planner (same names and argument meaning as the reference's
planning/planner_helper.py, re-written; file:line citations point into
/root/reference/car_racing/planning/planner_helper.py).

These stay on the host: they are a few dozen flops per control step.
"""
import numpy as np

from crx.hostprep import interp_clipped
from utils.constants import X_DIM


class AgentInfo:
    """Aggregates over the vehicles of interest (:269-276)."""

    __slots__ = ("max_delta_v", "min_delta_v", "max_s", "min_s", "max_vx", "min_vx")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)


def get_agent_range(s_agent, ey_agent, epsi_agent, length, width):
    """Axis-aligned extent of a rotated vehicle box in (s, ey) (:9-14)."""
    half_ey = 0.5 * length * np.sin(epsi_agent) + 0.5 * width * np.cos(epsi_agent)
    half_s = 0.5 * length * np.cos(epsi_agent) + 0.5 * width * np.sin(epsi_agent)
    return ey_agent + half_ey, ey_agent - half_ey, s_agent + half_s, s_agent - half_s


def ego_agent_overlap_checker(s_ego_min, s_ego_max, s_veh_min, s_veh_max, lap_length):
    """(:17-25).  The reference or-s three separation tests, one of which is always true for a
    positive lap length, so the result is False unless all three fail; kept as is."""
    apart = s_ego_max <= s_veh_min or s_ego_min >= s_veh_max
    apart_next = s_ego_max <= s_veh_min + lap_length or s_ego_min >= s_veh_max + lap_length
    apart_prev = s_ego_max + lap_length <= s_veh_min or s_ego_min + lap_length >= s_veh_max
    return not (apart or apart_next or apart_prev)


def _wrap_above(s, lap_length):
    while s > lap_length:
        s = s - lap_length
    return s


def check_ego_agent_distance(ego, agent, racing_game_param, lap_length):
    """Is `agent` close enough to matter? (:218-266): in front within
    safety_factor*length + prediction_factor*|dv|, or behind within one car length; both tests
    also with the agent / the ego shifted by one lap."""
    dv = abs(ego.xcurv[0] - agent.xcurv[0])
    s_a = _wrap_above(float(agent.xcurv[4]), lap_length)
    s_e = _wrap_above(float(ego.xcurv[4]), lap_length)
    ahead = racing_game_param.safety_factor * ego.param.length + racing_game_param.planning_prediction_factor * dv
    behind = 1.0 * ego.param.length
    return bool(
        (s_a - s_e <= ahead and s_a >= s_e)
        or (s_a + lap_length - s_e <= ahead and s_a + lap_length >= s_e)
        or (s_e - s_a <= behind and s_a <= s_e)
        or (s_e + lap_length - s_a <= behind and s_a <= s_e + lap_length)
    )


def get_agent_info(vehicles, sorted_vehicles, track):
    """(:177-201)."""
    vx = np.array([vehicles[n].xcurv[0] for n in sorted_vehicles], dtype=float)
    s = np.array([vehicles[n].xcurv[4] for n in sorted_vehicles], dtype=float)
    dv = np.abs(vehicles["ego"].xcurv[0] - vx)
    s = np.where(s <= 20, s + track.lap_length, s)  # next_lap_range = 20 (:178,:187-190)
    info = AgentInfo()
    info.min_vx, info.max_vx = vx.min(), vx.max()
    info.min_delta_v, info.max_delta_v = dv.min(), dv.max()
    info.min_s = _wrap_above(s.min(), track.lap_length)
    info.max_s = _wrap_above(s.max(), track.lap_length)
    return info


def bezier_control_points(num_veh, veh_info_list, max_delta_v, prediction_factor, track_width, lap_length,
                          veh_width, optimal_traj_xcurv, xcurv_ego):
    """Cubic Bezier control points in (s, ey), one curve per region (:43-135), on plain arrays.

    veh_info_list [num_veh,3] rows (s, max ey over the prediction, min ey) in the ITERATION order
    of vehicles_interest -- consumed as if sorted (quirk Q4, overtake_traj_planner.py:87-92)."""
    R = num_veh + 1
    cp = np.zeros((R, 4, 2))
    opt_s, opt_ey = optimal_traj_xcurv[:, 4], optimal_traj_xcurv[:, 5]
    s0 = xcurv_ego[4]
    s3 = s0 + prediction_factor * max_delta_v + 4  # :51-53
    if s0 > s3:  # :55-74 (start line between s0 and s3); unreachable for positive look-ahead, kept
        span = s3 + lap_length - s0
        s3 = s3 + lap_length
    else:
        span = s3 - s0
    cp[:, 0, 0] = s0
    cp[:, 1, 0] = span / 3.0 + s0
    cp[:, 2, 0] = 2.0 * span / 3.0 + s0
    cp[:, 3, 0] = s3
    cp[:, 0, 1] = xcurv_ego[5]  # :95 overrides :87-94
    for r in range(R):
        if r == 0:  # left of the left-most vehicle (:98-104)
            e = 0.8 * track_width - (-veh_info_list[r, 1] - 0.5 * veh_width) * 0.2
        elif r == num_veh:  # right of the right-most vehicle (:106-112)
            e = -0.8 * track_width + (veh_info_list[r - 1, 1] - 0.5 * veh_width) * 0.2
        else:  # between two vehicles (:113-119)
            e = 0.7 * (veh_info_list[r, 1] + 0.5 * veh_width) + 0.3 * (veh_info_list[r - 1, 1] - 0.5 * veh_width)
        cp[r, 1, 1] = cp[r, 2, 1] = e
    # end point rides the optimal trajectory (:121-134)
    s_end = s3 - lap_length if s3 >= lap_length else s3
    if s_end <= opt_s[0]:
        e3 = opt_ey[0]
    else:
        if s_end > opt_s[-1]:
            raise ValueError("A value in x_new is above the interpolation range.")  # interp1d bounds_error
        e3 = interp_clipped(opt_s, opt_ey, s_end)
    cp[:, 3, 1] = e3
    return cp


def get_bezier_control_points(vehicles_interest, veh_info_list, agent_info, racing_game_param, track,
                              optimal_traj_xcurv, sorted_vehicles, xcurv_ego):
    """Reference signature (:28-37)."""
    first = vehicles_interest[next(iter(vehicles_interest))]
    return bezier_control_points(
        len(vehicles_interest), np.asarray(veh_info_list, dtype=float), agent_info.max_delta_v,
        racing_game_param.planning_prediction_factor, track.width, track.lap_length, first.param.width,
        optimal_traj_xcurv, xcurv_ego)


def get_bezier_curve(bezier_control_point, t):
    """Point of the cubic at parameter t (:138-153); returns [s, ey]."""
    b = np.array([(1 - t) ** 3, 3 * t * (1 - t) ** 2, 3 * t ** 2 * (1 - t), t ** 3])
    p = bezier_control_point
    s = p[0, 0] * b[0] + p[1, 0] * b[1] + p[2, 0] * b[2] + p[3, 0] * b[3]
    e = p[0, 1] * b[0] + p[1, 1] * b[1] + p[2, 1] * b[2] + p[3, 1] * b[3]
    return [s, e]


def bezier_polylines(cp, N):
    """All regions' N+1 sample points (overtake_traj_planner.py:105-111): [R, N+1, 2]."""
    out = np.zeros((cp.shape[0], N + 1, 2))
    for j in range(N + 1):
        t = j * (1.0 / N)
        for r in range(cp.shape[0]):
            out[r, j, :] = get_bezier_curve(cp[r], t)
    return out


def get_traj_xglob(traj_xcurv, track):
    """(s, ey) -> (X, Y) for plotting (:204-215)."""
    n = np.size(traj_xcurv, 0)
    out = np.zeros((n, X_DIM))
    for i in range(n):
        s_i = _wrap_above(float(traj_xcurv[i, 4]), track.lap_length)
        out[i, 4], out[i, 5] = track.get_global_position(s_i, traj_xcurv[i, 5])
    return out


def sort_by_ey(names, ey_of):
    """The reference's partial 'sort' (overtake_traj_planner.py:70-76, quirk Q3): every new name is
    compared with the CURRENT FIRST element only."""
    order = []
    for n in names:
        if not order:
            order.append(n)
        elif ey_of(n) >= ey_of(order[0]):
            order.insert(0, n)
        elif ey_of(n) <= ey_of(order[0]):
            order.append(n)
    return order
