"""Vectorised host-side preparation of the arrays the crx C ABI consumes.

Everything here is closed-form NumPy over a leading batch axis.  Each function states which lines
of the reference (paths into /root/reference/car_racing) it restates; quirks Q1-Q9 of SURVEY.md
section 8a are reproduced on purpose.
"""
import numpy as np


def wrap_above(s, lap_length):
    """`while s > lap_length: s -= lap_length` (planning/overtake_traj_planner.py:291-292,216-217),
    vectorised; values <= lap_length (including negatives) are left alone."""
    s = np.asarray(s, dtype=float)
    k = np.ceil(s / lap_length) - 1.0
    k = np.maximum(k, 0.0)
    out = s - k * lap_length
    # guard the open/closed end exactly like the loop does (s == lap_length stays)
    out = np.where(out > lap_length, out - lap_length, out)
    return out


def planner_ey_bounds(x_wrapped, obs_s, obs_ey, n_veh, track_width, lap_length, N,
                      veh_length=0.4, veh_width=0.2, safety_margin=0.15, dt_ref=0.1):
    """Per-region bounds on ey_k, k = 0..N-1 (planning/overtake_traj_planner.py:277-324).

    x_wrapped [S,6]   the start-line-wrapped ego state `xcurv_ego` (window test, :296-300; quirk Q5)
    obs_s/ey  [S,V,N+1] predictions of the SORTED vehicles (obs_infos[sorted_vehicles[v]][4|5,:])
    n_veh     [S]
    Returns ey_lb [S,V+1,N], ey_ub [S,V+1].  Both neighbours impose ey >= ey_obs + W + margin
    (quirk Q2: the "right" neighbour uses the same inequality direction, :322).
    """
    x_wrapped = np.asarray(x_wrapped, dtype=float)
    S = x_wrapped.shape[0]
    V = obs_s.shape[1]
    ub = track_width - 0.5 * veh_width
    ey_lb = np.full((S, V + 1, N), -ub)
    ey_ub = np.full((S, V + 1), ub)
    k = np.arange(N)
    s_nom = x_wrapped[:, 4, None] + k[None, :] * dt_ref * x_wrapped[:, 0, None]  # [S,N]
    for v in range(V):
        os_ = wrap_above(obs_s[:, v, :N], lap_length)
        act = (s_nom >= os_ - veh_length - safety_margin) & (s_nom <= os_ + veh_length + safety_margin)
        act &= (v < np.asarray(n_veh))[:, None]
        need = obs_ey[:, v, :N] + veh_width + safety_margin
        cand = np.where(act, need, -np.inf)
        # vehicle v is the LEFT neighbour (sorted[r-1]) of region r = v+1 and the RIGHT neighbour
        # (sorted[r]) of region r = v
        for r in (v, v + 1):
            ey_lb[:, r, :] = np.maximum(ey_lb[:, r, :], cand)
    return ey_lb, ey_ub


def cbf_window(x_raw, obs_pred_s0, lap_length, safety_time=2.0):
    """Which obstacles enter the NLP and their lap offsets (control/control.py:499-523,538-540).

    x_raw [B,6]; obs_pred_s0 [B,V] = obs_traj[4,0] of every other vehicle.
    Returns keep [B,V] (bool), lap_off [B,V] = (num_cycle_ego - num_cycle_obs) * lap_length.
    `int()` truncates toward zero, as in the reference."""
    x_raw = np.asarray(x_raw, dtype=float)
    s0 = np.asarray(obs_pred_s0, dtype=float)
    margin = x_raw[:, 0] * safety_time
    nce = np.trunc(x_raw[:, 4] / lap_length)
    dist_ego = x_raw[:, 4] - nce * lap_length
    nco = np.trunc(s0 / lap_length)
    dist_obs = s0 - nco * lap_length
    keep = (dist_ego[:, None] > dist_obs - margin[:, None]) & (dist_ego[:, None] < dist_obs + margin[:, None])
    lap_off = (nce[:, None] - nco) * lap_length
    return keep, lap_off


def pack_obstacles(keep, obs_s, obs_ey, lap_off, n_obs_max, ego_s=None, dims=None):
    """Compact the kept obstacles to the front (dict order of the reference's obs_infos) and pad.  With `dims` [B, V, 2]
    (l_agent + l_obs, w_agent + w_obs of every vehicle, control.py:529-535) a fifth array follows: the dims of the kept ones."""
    B, V, L = obs_s.shape
    out_s = np.zeros((B, n_obs_max, L))
    out_e = np.zeros((B, n_obs_max, L))
    out_off = np.zeros((B, n_obs_max))
    n = np.zeros(B, dtype=np.int32)
    out_d = np.ones((B, n_obs_max, 2)) if dims is not None else None
    for b in range(B):
        idx = np.nonzero(keep[b])[0]
        if len(idx) > n_obs_max:
            # libcrx carries at most CRX_MAX_OBS obstacles per NLP (the reference has no limit, control.py:524-562): keep
            # the ones nearest to the ego at the first prediction step, in the reference's order, and say so
            import warnings

            if ego_s is None:
                raise ValueError("more obstacles in the window (%d) than n_obs_max (%d)" % (len(idx), n_obs_max))
            near = np.argsort(np.abs(obs_s[b, idx, 0] + lap_off[b, idx] - ego_s[b]), kind="stable")
            warnings.warn("%d obstacles pass the window test, libcrx keeps the %d nearest (CRX_MAX_OBS)" % (len(idx), n_obs_max))
            idx = np.sort(idx[near[:n_obs_max]])
        n[b] = len(idx)
        out_s[b, : len(idx)] = obs_s[b, idx]
        out_e[b, : len(idx)] = obs_ey[b, idx]
        out_off[b, : len(idx)] = lap_off[b, idx]
        if dims is not None:
            out_d[b, : len(idx)] = np.asarray(dims)[b, idx]
    if dims is not None:
        return out_s, out_e, out_off, n, out_d
    return out_s, out_e, out_off, n


def interp_clipped(xs, ys, x):
    """scipy interp1d(kind='linear') evaluated at x clipped to [xs[0], xs[-1]]
    (planning/overtake_traj_planner.py:331-332; control/control.py:373-378)."""
    x = np.clip(x, xs[0], xs[-1])
    hi = np.clip(np.searchsorted(xs, x, side="left"), 1, len(xs) - 1)
    lo = hi - 1
    slope = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo])
    return slope * (x - xs[lo]) + ys[lo]


def tracking_targets(x, traj_xcurv, N, dt_ref=0.1):
    """Per-stage targets of mpc_multi_agents (control/control.py:373-382):
    xt_i = [vx0,0,0,0,0, f_traj(clip(s0 + 0.1*i*vx0))]."""
    x = np.asarray(x, dtype=float)
    xt = np.zeros((N + 1, 6))
    s = x[4] + x[0] * dt_ref * np.arange(N + 1)
    s = np.where(s < traj_xcurv[0, 4], traj_xcurv[0, 4], s)
    s = np.where(s >= traj_xcurv[-1, 4], traj_xcurv[-1, 4], s)
    xt[:, 0] = x[0]
    xt[:, 5] = interp_clipped(traj_xcurv[:, 4], traj_xcurv[:, 5], s)
    return xt


def tracking_targets_batch(x, traj_xcurv, N, dt_ref=0.1):
    """tracking_targets for a batch: x [B,6], traj_xcurv [B,M,6] -> xt [B,N+1,6]."""
    x = np.asarray(x, dtype=float)
    return np.stack([tracking_targets(x[b], traj_xcurv[b], N, dt_ref) for b in range(x.shape[0])])
