"""Device-resident entry points: torch CUDA(HIP) tensors in, torch tensors out, no host staging.

torch is used for device memory, streams and torch.distributed only; the solve itself is
libcrx's `*_dev` C ABI called with raw device pointers on torch's current stream.
"""
import ctypes as C

import torch

from . import _state, binding, lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t, dtype, shape, name):
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
        raise ValueError("%s: expected contiguous cuda %s tensor of shape %s, got %s %s %s" % (
            name, dtype, tuple(shape), t.device, t.dtype, tuple(t.shape)))
    # the *_dev entry points launch on the device libcrx was initialised on: a tensor living elsewhere would be
    # dereferenced on the wrong GPU
    dev = _state["device"]
    if dev is not None and t.device.index != dev:
        raise ValueError("%s lives on %s but libcrx was initialised on cuda:%d (crx.init(device))" % (name, t.device, dev))
    return t


def _call(name, *args):
    binding()
    L = lib()
    fn = getattr(L, name)
    fn.restype = C.c_int
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError("%s failed: rc=%d %s" % (name, rc, (L.crx_last_error() or b"").decode()))


def _stream():
    binding()
    return C.c_void_p(torch.cuda.current_stream(torch.device("cuda", _state["device"])).cuda_stream)


def new_streams(n, device=None):
    """crx_streams_create: n HIP streams of which the first `n_concurrent` were MEASURED to sit on pairwise different hardware
    queues (torch's own stream pool gives no such guarantee: include/crx.h "Streams"), wrapped as torch.cuda.ExternalStream.
    Returns (streams, n_concurrent).  They live as long as the process."""
    binding()
    arr = (C.c_void_p * n)()
    nc = C.c_int(0)
    _call("crx_streams_create", C.c_int(n), arr, C.byref(nc))
    dev = torch.device("cuda", _state["device"]) if device is None else torch.device(device)
    return [torch.cuda.ExternalStream(int(arr[i]), device=dev) for i in range(n)], int(nc.value)


class Timer:
    """crx_timer_* (include/crx.h): device time of what is enqueued between begin() and end() on torch's current stream."""

    def __init__(self):
        binding()
        self.h = C.c_void_p(0)
        _call("crx_timer_create", C.byref(self.h))

    def begin(self):
        _call("crx_timer_begin", self.h, _stream())

    def end(self):
        _call("crx_timer_end", self.h, _stream())

    def ms(self):
        return float(lib().crx_timer_ms(self.h))

    def __del__(self):
        try:
            if self.h:
                lib().crx_timer_destroy(self.h)
        except Exception:
            pass


class CbfWorkspace:
    """Pre-allocated outputs for repeated cbf_solve_dev calls of one shape."""

    def __init__(self, desc, batch, device):
        N, V = desc.N, max(desc.n_obs_max, 1)
        f64 = dict(dtype=torch.float64, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.X = torch.empty((batch, N + 1, 6), **f64)
        self.U = torch.empty((batch, N, 2), **f64)
        self.sigma = torch.empty((batch, V, N + 1), **f64)
        self.cost = torch.empty(batch, **f64)
        self.kkt = torch.empty(batch, **f64)
        self.status = torch.empty(batch, **i32)
        self.iters = torch.zeros(batch, **i32)


def longest_first(iters, active=None, out=None):
    """crx_order_longest_first_dev: dispatch order for the next launch from the iteration counts of the previous one (int32 [batch]
    on the device): the problems that took longest start first, masked-out ones (active == 0) last (include/crx.h, "Dispatch
    order").  Stable: equal counts keep index order."""
    Bn = iters.shape[0]
    _chk(iters, torch.int32, (Bn,), "iters")
    if active is not None:
        _chk(active, torch.int32, (Bn,), "active")
    order = out if out is not None else torch.empty(Bn, dtype=torch.int32, device=iters.device)
    _chk(order, torch.int32, (Bn,), "order")
    _call("crx_order_longest_first_dev", C.c_int(Bn), _ptr(iters), _ptr(active), _ptr(order), _stream())
    return order


def cbf_order_dev(desc, x0, xt, obs_s, obs_ey, lap_off, n_obs, obs_dims=None, active=None, out=None):
    """crx_cbf_order_dev: dispatch order of a CBF-NLP launch with no previous solve to go by, from the arguments of cbf_solve_dev --
    cars that start inside a safety ellipse first, then those whose un-steered path enters one, then the rest."""
    N, V, Bn = desc.N, desc.n_obs_max, x0.shape[0]
    _chk(x0, torch.float64, (Bn, 6), "x0")
    _chk(xt, torch.float64, (Bn, N + 1, 6) if desc.per_stage_target else (Bn, 6), "xt")
    _chk(obs_s, torch.float64, (Bn, V, N + 1), "obs_s")
    _chk(obs_ey, torch.float64, (Bn, V, N + 1), "obs_ey")
    _chk(lap_off, torch.float64, (Bn, V), "lap_off")
    _chk(n_obs, torch.int32, (Bn,), "n_obs")
    if obs_dims is not None:
        _chk(obs_dims, torch.float64, (Bn, V, 2), "obs_dims")
    if active is not None:
        _chk(active, torch.int32, (Bn,), "active")
    order = out if out is not None else torch.empty(Bn, dtype=torch.int32, device=x0.device)
    _chk(order, torch.int32, (Bn,), "order")
    _call("crx_cbf_order_dev", C.byref(desc), C.c_int(Bn), _ptr(active), _ptr(x0), _ptr(xt), _ptr(obs_s), _ptr(obs_ey), _ptr(lap_off), _ptr(n_obs),
          _ptr(obs_dims), _ptr(order), _stream())
    return order


def cbf_solve_dev(desc, x0, xt, obs_s, obs_ey, lap_off, n_obs, ws=None, active=None, obs_dims=None, order=None):
    """crx_cbf_solve_ordered_dev, the superset entry point: `active` (int32 [batch], 0 = leave the problem alone), `obs_dims`
    ([batch, n_obs_max, 2]: l_agent + l_obs, w_agent + w_obs per obstacle slot), `order` (int32 [batch] permutation: workgroup i
    solves problem order[i]; see longest_first) -- each optional."""
    N, V, B = desc.N, desc.n_obs_max, x0.shape[0]
    _chk(x0, torch.float64, (B, 6), "x0")
    _chk(xt, torch.float64, (B, N + 1, 6) if desc.per_stage_target else (B, 6), "xt")
    _chk(obs_s, torch.float64, (B, V, N + 1), "obs_s")
    _chk(obs_ey, torch.float64, (B, V, N + 1), "obs_ey")
    _chk(lap_off, torch.float64, (B, V), "lap_off")
    _chk(n_obs, torch.int32, (B,), "n_obs")
    ws = ws or CbfWorkspace(desc, B, x0.device)
    if active is not None:
        _chk(active, torch.int32, (B,), "active")
    if obs_dims is not None:
        _chk(obs_dims, torch.float64, (B, V, 2), "obs_dims")
    if order is not None:
        _chk(order, torch.int32, (B,), "order")
    _call("crx_cbf_solve_ordered_dev", C.byref(desc), C.c_int(B), _ptr(active) if active is not None else None,
          _ptr(order) if order is not None else None, _ptr(x0), _ptr(xt),
          _ptr(obs_s), _ptr(obs_ey), _ptr(lap_off), _ptr(n_obs), _ptr(obs_dims), _ptr(ws.X), _ptr(ws.U), _ptr(ws.sigma), _ptr(ws.cost),
          _ptr(ws.status), _ptr(ws.kkt), _ptr(ws.iters), _stream())
    return ws


class PlannerWorkspace:
    def __init__(self, desc, batch, device):
        N = desc.N
        f64 = dict(dtype=torch.float64, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.X = torch.empty((batch, N + 1, 6), **f64)
        self.U = torch.empty((batch, N, 2), **f64)
        self.cost = torch.empty(batch, **f64)
        self.kkt = torch.empty(batch, **f64)
        self.status = torch.empty(batch, **i32)
        self.iters = torch.empty(batch, **i32)


def planner_solve_dev(desc, x0, bez_s, bez_ey, ey_lb, ey_ub, ws=None):
    N, B = desc.N, x0.shape[0]
    _chk(x0, torch.float64, (B, 6), "x0")
    _chk(bez_s, torch.float64, (B, N + 1), "bez_s")
    _chk(bez_ey, torch.float64, (B, N + 1), "bez_ey")
    _chk(ey_lb, torch.float64, (B, N), "ey_lb")
    _chk(ey_ub, torch.float64, (B,), "ey_ub")
    ws = ws or PlannerWorkspace(desc, B, x0.device)
    _call("crx_planner_solve_dev", C.byref(desc), C.c_int(B), _ptr(x0), _ptr(bez_s), _ptr(bez_ey), _ptr(ey_lb),
          _ptr(ey_ub), _ptr(ws.X), _ptr(ws.U), _ptr(ws.cost), _ptr(ws.status), _ptr(ws.kkt), _ptr(ws.iters),
          _stream())
    return ws


class SelectWorkspace:
    def __init__(self, desc, n_scen, device):
        N, R = desc.N, desc.n_veh_max + 1
        self.flag = torch.empty(n_scen, dtype=torch.int32, device=device)
        self.sel_cost = torch.empty((n_scen, R), dtype=torch.float64, device=device)
        self.best_X = torch.empty((n_scen, N + 1, 6), dtype=torch.float64, device=device)


def select_dev(desc, n_veh, X, obs_s, obs_ey, old_flag, ws=None):
    N, V, S = desc.N, desc.n_veh_max, n_veh.shape[0]
    _chk(n_veh, torch.int32, (S,), "n_veh")
    _chk(X, torch.float64, (S, V + 1, N + 1, 6), "X")
    _chk(obs_s, torch.float64, (S, V, N + 1), "obs_s")
    _chk(obs_ey, torch.float64, (S, V, N + 1), "obs_ey")
    _chk(old_flag, torch.int32, (S,), "old_flag")
    ws = ws or SelectWorkspace(desc, S, X.device)
    _call("crx_select_dev", C.byref(desc), C.c_int(S), _ptr(n_veh), _ptr(X), _ptr(obs_s), _ptr(obs_ey),
          _ptr(old_flag), _ptr(ws.flag), _ptr(ws.sel_cost), _ptr(ws.best_X), _stream())
    return ws


class LmpcWorkspace:
    def __init__(self, desc, batch, device):
        N, M = desc.N, desc.n_ss_max
        f64 = dict(dtype=torch.float64, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.X = torch.empty((batch, N + 1, 6), **f64)
        self.U = torch.empty((batch, N, 2), **f64)
        self.lam = torch.empty((batch, M), **f64)
        self.cost = torch.empty(batch, **f64)
        self.kkt = torch.empty(batch, **f64)
        self.status = torch.empty(batch, **i32)
        self.iters = torch.zeros(batch, **i32)


def lmpc_solve_dev(desc, x0, u_old, A, B, Cm, ss, qfun, n_ss, ws=None, active=None, order=None):
    """crx_lmpc_solve_ordered_dev (`active`, int32 [batch], 0 = leave alone; `order`, int32 [batch] dispatch permutation; both optional).  n_ss must satisfy 1 <= n_ss <= desc.n_ss_max (checked here on the host copy the
    caller keeps; the kernel indexes LDS with it)."""
    N, M, Bn = desc.N, desc.n_ss_max, x0.shape[0]
    _chk(x0, torch.float64, (Bn, 6), "x0")
    _chk(u_old, torch.float64, (Bn, 2), "u_old")
    _chk(A, torch.float64, (Bn, N, 36), "A")
    _chk(B, torch.float64, (Bn, N, 12), "B")
    _chk(Cm, torch.float64, (Bn, N, 6), "C")
    _chk(ss, torch.float64, (Bn, 6, M), "ss")
    _chk(qfun, torch.float64, (Bn, M), "qfun")
    _chk(n_ss, torch.int32, (Bn,), "n_ss")
    ws = ws or LmpcWorkspace(desc, Bn, x0.device)
    if active is not None:
        _chk(active, torch.int32, (Bn,), "active")
    if order is not None:
        _chk(order, torch.int32, (Bn,), "order")
    _call("crx_lmpc_solve_ordered_dev", C.byref(desc), C.c_int(Bn), _ptr(active) if active is not None else None,
          _ptr(order) if order is not None else None, _ptr(x0), _ptr(u_old),
          _ptr(A), _ptr(B), _ptr(Cm), _ptr(ss), _ptr(qfun), _ptr(n_ss), _ptr(ws.X), _ptr(ws.U), _ptr(ws.lam), _ptr(ws.cost),
          _ptr(ws.status), _ptr(ws.kkt), _ptr(ws.iters), _stream())
    return ws


class PrepWorkspace:
    """Outputs of planner_prep_dev = inputs of planner_solve_dev (same layout)."""

    def __init__(self, desc, n_scen, device):
        N, R = desc.N, desc.n_veh_max + 1
        f64 = dict(dtype=torch.float64, device=device)
        self.x0 = torch.empty((n_scen * R, 6), **f64)
        self.bez_s = torch.empty((n_scen * R, N + 1), **f64)
        self.bez_ey = torch.empty((n_scen * R, N + 1), **f64)
        self.ey_lb = torch.empty((n_scen * R, N), **f64)
        self.ey_ub = torch.empty((n_scen * R,), **f64)


def planner_prep_dev(desc, x_wrapped, x_raw, n_veh, veh_info, max_dv, obs_s, obs_ey, opt_s, opt_ey, ws=None):
    N, V, T, S = desc.N, desc.n_veh_max, desc.n_opt, x_wrapped.shape[0]
    _chk(x_wrapped, torch.float64, (S, 6), "x_wrapped")
    _chk(x_raw, torch.float64, (S, 6), "x_raw")
    _chk(n_veh, torch.int32, (S,), "n_veh")
    _chk(veh_info, torch.float64, (S, V, 3), "veh_info")
    _chk(max_dv, torch.float64, (S,), "max_dv")
    _chk(obs_s, torch.float64, (S, V, N + 1), "obs_s")
    _chk(obs_ey, torch.float64, (S, V, N + 1), "obs_ey")
    _chk(opt_s, torch.float64, (T,), "opt_s")
    _chk(opt_ey, torch.float64, (T,), "opt_ey")
    ws = ws or PrepWorkspace(desc, S, x_wrapped.device)
    _call("crx_planner_prep_dev", C.byref(desc), C.c_int(S), _ptr(x_wrapped), _ptr(x_raw), _ptr(n_veh), _ptr(veh_info),
          _ptr(max_dv), _ptr(obs_s), _ptr(obs_ey), _ptr(opt_s), _ptr(opt_ey), _ptr(ws.x0), _ptr(ws.bez_s),
          _ptr(ws.bez_ey), _ptr(ws.ey_lb), _ptr(ws.ey_ub), _stream())
    return ws


def plant_step_dev(desc, track, xglob, xcurv, u, xglob_next=None, xcurv_next=None):
    """crx_plant_step_dev; returns (xglob_next, xcurv_next) (allocated unless given)."""
    Bn = xglob.shape[0]
    _chk(track, torch.float64, (desc.n_seg, 6), "track")
    _chk(xglob, torch.float64, (Bn, 6), "xglob")
    _chk(xcurv, torch.float64, (Bn, 6), "xcurv")
    _chk(u, torch.float64, (Bn, 2), "u")
    xglob_next = torch.empty_like(xglob) if xglob_next is None else xglob_next
    xcurv_next = torch.empty_like(xcurv) if xcurv_next is None else xcurv_next
    _call("crx_plant_step_dev", C.byref(desc), C.c_int(Bn), _ptr(track), _ptr(xglob), _ptr(xcurv), _ptr(u),
          _ptr(xglob_next), _ptr(xcurv_next), _stream())
    return xglob_next, xcurv_next


def cbf_prep_dev(N, lap_length, t, dt, xcurv, car_s0, car_v, car_ey, obs_s, obs_ey, lap_off, n_obs, safety_time=2.0):
    """crx_cbf_prep_dev: obstacle inputs of cbf_solve_dev for scripted cars, written into the given tensors."""
    Bn, V = car_s0.shape
    _chk(xcurv, torch.float64, (Bn, 6), "xcurv")
    for name, a in (("car_s0", car_s0), ("car_v", car_v), ("car_ey", car_ey), ("lap_off", lap_off)):
        _chk(a, torch.float64, (Bn, V), name)
    _chk(obs_s, torch.float64, (Bn, V, N + 1), "obs_s")
    _chk(obs_ey, torch.float64, (Bn, V, N + 1), "obs_ey")
    _chk(n_obs, torch.int32, (Bn,), "n_obs")
    _call("crx_cbf_prep_dev", C.c_int(N), C.c_int(V), C.c_double(lap_length), C.c_double(t), C.c_double(dt),
          C.c_double(safety_time), C.c_int(Bn), _ptr(xcurv), _ptr(car_s0), _ptr(car_v), _ptr(car_ey), _ptr(obs_s),
          _ptr(obs_ey), _ptr(lap_off), _ptr(n_obs), _stream())


def plant_step_wrap_dev(desc, track, xglob, xcurv, u, u_stride, xglob_next, xcurv_next, laps, noise_z=None):
    """crx_plant_step_wrap_dev: plant step + lap bookkeeping; `u` may be a strided view's base tensor (u_stride doubles
    between vehicles, e.g. the U output of cbf_solve_dev with u_stride = 2 N).  noise_z [B,3]: standard-normal draws for the
    reference's bounded process noise (utils/base.py:929-939; crx_plant_step_noise_dev), None = zero noise."""
    Bn = xglob.shape[0]
    _chk(track, torch.float64, (desc.n_seg, 6), "track")
    _chk(xglob, torch.float64, (Bn, 6), "xglob")
    _chk(xcurv, torch.float64, (Bn, 6), "xcurv")
    _chk(xglob_next, torch.float64, (Bn, 6), "xglob_next")
    _chk(xcurv_next, torch.float64, (Bn, 6), "xcurv_next")
    _chk(laps, torch.int32, (Bn,), "laps")
    if not u.is_cuda or u.dtype != torch.float64 or not u.is_contiguous() or u.numel() < (Bn - 1) * u_stride + 2:
        raise ValueError("u: expected a contiguous cuda float64 tensor holding %d inputs at stride %d" % (Bn, u_stride))
    if noise_z is not None:
        _chk(noise_z, torch.float64, (Bn, 3), "noise_z")
    _call("crx_plant_step_noise_dev", C.byref(desc), C.c_int(Bn), _ptr(track), _ptr(xglob), _ptr(xcurv), _ptr(u),
          C.c_int(u_stride), _ptr(noise_z), _ptr(xglob_next), _ptr(xcurv_next), _ptr(laps), _stream())


class LmpcPrepWorkspace:
    """Outputs of lmpc_prep_dev = the model and safe-set inputs of lmpc_solve_dev (same layout)."""

    def __init__(self, desc, batch, device):
        N, M = desc.N, desc.n_ss_per_lap * desc.n_ss_laps
        f64 = dict(dtype=torch.float64, device=device)
        # A, B, C are in/out: the kernel leaves the three regression rows of a singular stage (status 1) and everything of a
        # masked-out race untouched, so they start defined (zeros), never as uninitialised memory that could reach the QP
        self.A = torch.zeros((batch, N, 36), **f64)
        self.B = torch.zeros((batch, N, 12), **f64)
        self.C = torch.zeros((batch, N, 6), **f64)
        self.ss = torch.zeros((batch, 6, M), **f64)
        self.qfun = torch.zeros((batch, M), **f64)
        self.status = torch.zeros(batch, dtype=torch.int32, device=device)


def lmpc_prep_dev(desc, ss_xcurv, u_ss, qfun, time_ss, it, x, lin_points, lin_input, track, from_plan, ws=None, active=None):
    """crx_lmpc_prep_dev: the N stage models and the safe-set selection of every race (of the races with active != 0:
    crx_lmpc_prep_masked_dev)."""
    N, P, L, Bn = desc.N, desc.n_points, desc.n_laps, x.shape[0]
    _chk(ss_xcurv, torch.float64, (Bn, L, P, 6), "ss_xcurv")
    _chk(u_ss, torch.float64, (Bn, L, P, 2), "u_ss")
    _chk(qfun, torch.float64, (Bn, L, P), "qfun")
    _chk(time_ss, torch.int32, (Bn, L), "time_ss")
    _chk(it, torch.int32, (Bn,), "iter")
    _chk(x, torch.float64, (Bn, 6), "x")
    _chk(lin_points, torch.float64, (Bn, N + 1, 6), "lin_points")
    _chk(lin_input, torch.float64, (Bn, N, 2), "lin_input")
    _chk(track, torch.float64, (desc.n_seg, 6), "track")
    ws = ws or LmpcPrepWorkspace(desc, Bn, x.device)
    if active is not None:
        _chk(active, torch.int32, (Bn,), "active")
    _call("crx_lmpc_prep_masked_dev", C.byref(desc), C.c_int(Bn), _ptr(active) if active is not None else None, _ptr(ss_xcurv), _ptr(u_ss),
          _ptr(qfun), _ptr(time_ss), _ptr(it), _ptr(x),
          _ptr(lin_points), _ptr(lin_input), C.c_int(int(bool(from_plan))), _ptr(track), _ptr(ws.A), _ptr(ws.B), _ptr(ws.C),
          _ptr(ws.ss), _ptr(ws.qfun), _ptr(ws.status), _stream())
    return ws


def lmpc_addpoint_dev(desc, ss_xcurv, u_ss, time_ss, it, step, x, u, u_stride):
    """crx_lmpc_addpoint_dev: LMPCRacingGame.add_point for every race (in place)."""
    Bn, P, L = x.shape[0], desc.n_points, desc.n_laps
    _chk(ss_xcurv, torch.float64, (Bn, L, P, 6), "ss_xcurv")
    _chk(u_ss, torch.float64, (Bn, L, P, 2), "u_ss")
    _chk(time_ss, torch.int32, (Bn, L), "time_ss")
    _chk(it, torch.int32, (Bn,), "iter")
    _chk(step, torch.int32, (Bn,), "step")
    _chk(x, torch.float64, (Bn, 6), "x")
    if not u.is_cuda or u.dtype != torch.float64 or not u.is_contiguous() or u.numel() < (Bn - 1) * u_stride + 2:
        raise ValueError("u: expected a contiguous cuda float64 tensor holding %d inputs at stride %d" % (Bn, u_stride))
    _call("crx_lmpc_addpoint_dev", C.byref(desc), C.c_int(Bn), _ptr(ss_xcurv), _ptr(u_ss), _ptr(time_ss), _ptr(it), _ptr(step),
          _ptr(x), _ptr(u), C.c_int(u_stride), _stream())


def lmpc_addtraj_dev(desc, crossed, log_x, log_u, n_log, ss_xcurv, u_ss, qfun, time_ss, it, step, x, status):
    """crx_lmpc_addtraj_dev: LMPCRacingGame.add_trajectory for the races with crossed != 0 (everything in place)."""
    Bn, P, L = x.shape[0], desc.n_points, desc.n_laps
    _chk(crossed, torch.int32, (Bn,), "crossed")
    _chk(log_x, torch.float64, (Bn, P, 6), "log_x")
    _chk(log_u, torch.float64, (Bn, P, 2), "log_u")
    _chk(n_log, torch.int32, (Bn,), "n_log")
    _chk(ss_xcurv, torch.float64, (Bn, L, P, 6), "ss_xcurv")
    _chk(u_ss, torch.float64, (Bn, L, P, 2), "u_ss")
    _chk(qfun, torch.float64, (Bn, L, P), "qfun")
    _chk(time_ss, torch.int32, (Bn, L), "time_ss")
    _chk(it, torch.int32, (Bn,), "iter")
    _chk(step, torch.int32, (Bn,), "step")
    _chk(x, torch.float64, (Bn, 6), "x")
    _chk(status, torch.int32, (Bn,), "status")
    _call("crx_lmpc_addtraj_dev", C.byref(desc), C.c_int(Bn), _ptr(crossed), _ptr(log_x), _ptr(log_u), _ptr(n_log), _ptr(ss_xcurv),
          _ptr(u_ss), _ptr(qfun), _ptr(time_ss), _ptr(it), _ptr(step), _ptr(x), _ptr(status), _stream())


class SceneWorkspace:
    def __init__(self, desc, n_scen, device):
        N1, V = desc.N + 1, desc.n_veh_max
        f64 = dict(dtype=torch.float64, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.n_veh, self.overflow = torch.empty(n_scen, **i32), torch.empty(n_scen, **i32)
        self.order = torch.empty((n_scen, V), **i32)
        self.veh_info = torch.empty((n_scen, V, 3), **f64)
        self.max_dv = torch.empty(n_scen, **f64)
        self.obs_s, self.obs_ey = torch.empty((n_scen, V, N1), **f64), torch.empty((n_scen, V, N1), **f64)


def planner_scene_dev(desc, ego_xcurv, n_all, veh_xcurv, pred_s, pred_ey, ws=None):
    """crx_planner_scene_dev: interest test, partial sort, veh_infos, max_delta_v, predictions in sorted order."""
    N1, VA, S = desc.N + 1, desc.n_all_max, ego_xcurv.shape[0]
    _chk(ego_xcurv, torch.float64, (S, 6), "ego_xcurv")
    _chk(n_all, torch.int32, (S,), "n_all")
    _chk(veh_xcurv, torch.float64, (S, VA, 6), "veh_xcurv")
    _chk(pred_s, torch.float64, (S, VA, N1), "pred_s")
    _chk(pred_ey, torch.float64, (S, VA, N1), "pred_ey")
    ws = ws or SceneWorkspace(desc, S, ego_xcurv.device)
    _call("crx_planner_scene_dev", C.byref(desc), C.c_int(S), _ptr(ego_xcurv), _ptr(n_all), _ptr(veh_xcurv), _ptr(pred_s), _ptr(pred_ey),
          _ptr(ws.n_veh), _ptr(ws.overflow), _ptr(ws.order), _ptr(ws.veh_info), _ptr(ws.max_dv), _ptr(ws.obs_s), _ptr(ws.obs_ey), _stream())
    return ws


def planner_plan_dev(desc, sdesc, x0, bez_s, bez_ey, ey_lb, ey_ub, n_veh, obs_s, obs_ey, old_flag, ws, sws, active=None):
    """crx_planner_plan_dev: all region QPs of every scenario + the selection, on one stream (with `active`, int32
    [n_scen]: crx_planner_plan_masked_dev, the QPs of the scenarios with 0 are skipped)."""
    S, N, V = n_veh.shape[0], desc.N, sdesc.n_veh_max
    R = V + 1
    _chk(x0, torch.float64, (S * R, 6), "x0")
    _chk(bez_s, torch.float64, (S * R, N + 1), "bez_s")
    _chk(bez_ey, torch.float64, (S * R, N + 1), "bez_ey")
    _chk(ey_lb, torch.float64, (S * R, N), "ey_lb")
    _chk(ey_ub, torch.float64, (S * R,), "ey_ub")
    _chk(n_veh, torch.int32, (S,), "n_veh")
    _chk(obs_s, torch.float64, (S, V, N + 1), "obs_s")
    _chk(obs_ey, torch.float64, (S, V, N + 1), "obs_ey")
    _chk(old_flag, torch.int32, (S,), "old_flag")
    _chk(ws.X, torch.float64, (S * R, N + 1, 6), "ws.X")
    _chk(ws.U, torch.float64, (S * R, N, 2), "ws.U")
    for name in ("cost", "kkt"):
        _chk(getattr(ws, name), torch.float64, (S * R,), "ws." + name)
    for name in ("status", "iters"):
        _chk(getattr(ws, name), torch.int32, (S * R,), "ws." + name)
    _chk(sws.flag, torch.int32, (S,), "sws.flag")
    _chk(sws.sel_cost, torch.float64, (S, R), "sws.sel_cost")
    _chk(sws.best_X, torch.float64, (S, N + 1, 6), "sws.best_X")
    if active is not None:
        _chk(active, torch.int32, (S,), "active")
    _call("crx_planner_plan_masked_dev", C.byref(desc), C.byref(sdesc), C.c_int(S), _ptr(active), _ptr(x0), _ptr(bez_s), _ptr(bez_ey), _ptr(ey_lb), _ptr(ey_ub),
          _ptr(n_veh), _ptr(obs_s), _ptr(obs_ey), _ptr(old_flag), _ptr(ws.X), _ptr(ws.U), _ptr(ws.cost), _ptr(ws.status), _ptr(ws.kkt),
          _ptr(ws.iters), _ptr(sws.flag), _ptr(sws.sel_cost), _ptr(sws.best_X), _stream())


def track_prep_dev(N, V, lap_length, x, n_veh, obs_s_in, obs_ey_in, traj, xt, obs_s, obs_ey, lap_off, n_obs, safety_time=2.0, dt_ref=0.1):
    """crx_track_prep_dev: per-stage targets and obstacle arrays of the tracking NLP from the planner's outputs."""
    Bn = x.shape[0]
    _chk(x, torch.float64, (Bn, 6), "x")
    _chk(n_veh, torch.int32, (Bn,), "n_veh")
    for name, a in (("obs_s_in", obs_s_in), ("obs_ey_in", obs_ey_in), ("obs_s", obs_s), ("obs_ey", obs_ey)):
        _chk(a, torch.float64, (Bn, V, N + 1), name)
    _chk(traj, torch.float64, (Bn, N + 1, 6), "traj")
    _chk(xt, torch.float64, (Bn, N + 1, 6), "xt")
    _chk(lap_off, torch.float64, (Bn, V), "lap_off")
    _chk(n_obs, torch.int32, (Bn,), "n_obs")
    _call("crx_track_prep_dev", C.c_int(N), C.c_int(V), C.c_double(lap_length), C.c_double(safety_time), C.c_double(dt_ref), C.c_int(Bn),
          _ptr(x), _ptr(n_veh), _ptr(obs_s_in), _ptr(obs_ey_in), _ptr(traj), _ptr(xt), _ptr(obs_s), _ptr(obs_ey), _ptr(lap_off), _ptr(n_obs),
          _stream())


# ---- device-resident racing-game loop: bookkeeping between the solver launches (include/crx.h crx_game_*) ----------------------
def game_traffic_dev(N, lap_length, t, dt, car_s0, car_v, car_ey, veh_xcurv, pred_s, pred_ey):
    Bn, VA = car_s0.shape
    for name, a in (("car_s0", car_s0), ("car_v", car_v), ("car_ey", car_ey)):
        _chk(a, torch.float64, (Bn, VA), name)
    _chk(veh_xcurv, torch.float64, (Bn, VA, 6), "veh_xcurv")
    _chk(pred_s, torch.float64, (Bn, VA, N + 1), "pred_s")
    _chk(pred_ey, torch.float64, (Bn, VA, N + 1), "pred_ey")
    _call("crx_game_traffic_dev", C.c_int(N), C.c_int(Bn), C.c_int(VA), C.c_double(lap_length), C.c_double(t), C.c_double(dt), _ptr(car_s0),
          _ptr(car_v), _ptr(car_ey), _ptr(veh_xcurv), _ptr(pred_s), _ptr(pred_ey), _stream())


def game_masks_dev(n_veh, m_overtake, m_lmpc, overflow=None, overflow_seen=None):
    Bn = n_veh.shape[0]
    for name, a in (("n_veh", n_veh), ("m_overtake", m_overtake), ("m_lmpc", m_lmpc)) + ((("overflow", overflow), ("overflow_seen", overflow_seen)) if overflow is not None else ()):
        _chk(a, torch.int32, (Bn,), name)
    _call("crx_game_masks_dev", C.c_int(Bn), _ptr(n_veh), _ptr(overflow), _ptr(m_overtake), _ptr(m_lmpc), _ptr(overflow_seen), _stream())


def game_commit_dev(N, Np, overtake, U_track, X_lmpc, U_lmpc, flag, u, u_old, u_prev, lin_points, lin_input, step_no, addpoint_step, old_flag):
    Bn = u.shape[0]
    _chk(X_lmpc, torch.float64, (Bn, N + 1, 6), "X_lmpc")
    _chk(U_lmpc, torch.float64, (Bn, N, 2), "U_lmpc")
    for name, a in (("u", u), ("u_old", u_old), ("u_prev", u_prev)):
        _chk(a, torch.float64, (Bn, 2), name)
    _chk(lin_points, torch.float64, (Bn, N + 1, 6), "lin_points")
    _chk(lin_input, torch.float64, (Bn, N, 2), "lin_input")
    _chk(step_no, torch.int32, (Bn,), "step_no")
    _chk(addpoint_step, torch.int32, (Bn,), "addpoint_step")
    if overtake is not None:
        _chk(overtake, torch.int32, (Bn,), "overtake")
        _chk(U_track, torch.float64, (Bn, Np, 2), "U_track")
        _chk(flag, torch.int32, (Bn,), "flag")
        _chk(old_flag, torch.int32, (Bn,), "old_flag")
    _call("crx_game_commit_dev", C.c_int(N), C.c_int(Np), C.c_int(Bn), _ptr(overtake), _ptr(U_track), _ptr(X_lmpc), _ptr(U_lmpc), _ptr(flag),
          _ptr(u), _ptr(u_old), _ptr(u_prev), _ptr(lin_points), _ptr(lin_input), _ptr(step_no), _ptr(addpoint_step), _ptr(old_flag), _stream())


def game_log_dev(lap_length, xcurv, u, laps, laps_prev, log_x, log_u, n_log, crossed):
    Bn, P = log_x.shape[0], log_x.shape[1]
    _chk(xcurv, torch.float64, (Bn, 6), "xcurv")
    _chk(u, torch.float64, (Bn, 2), "u")
    for name, a in (("laps", laps), ("laps_prev", laps_prev), ("n_log", n_log), ("crossed", crossed)):
        _chk(a, torch.int32, (Bn,), name)
    _chk(log_x, torch.float64, (Bn, P, 6), "log_x")
    _chk(log_u, torch.float64, (Bn, P, 2), "log_u")
    _call("crx_game_log_dev", C.c_int(Bn), C.c_int(P), C.c_double(lap_length), _ptr(xcurv), _ptr(u), _ptr(laps), _ptr(laps_prev), _ptr(log_x),
          _ptr(log_u), _ptr(n_log), _ptr(crossed), _stream())
