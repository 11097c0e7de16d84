"""ctypes mirror of include/crx.h (struct layouts + argument marshalling).

Pure declarations: no library is loaded here.  `Binding` wraps any CDLL that exports the crx entry
points under a given symbol prefix, so the same marshalling code drives libcrx (prefix "crx_") and,
in tests only, the CPU oracle (prefix "crx_oracle_").
"""
import ctypes as C

import numpy as np

CRX_MAX_N = 24
CRX_MAX_OBS = 6     # obstacles per MPC-CBF NLP
CRX_MAX_VEH = 6     # vehicles of interest per planner scenario (= CRX_MAX_OBS)

CRX_CONVERGED, CRX_MAX_ITER, CRX_INFEASIBLE, CRX_RESTORED, CRX_SKIPPED, CRX_STALLED = 0, 1, 2, 3, 4, 5


class IpmOpts(C.Structure):
    _fields_ = [
        ("tol", C.c_double),
        ("max_iter", C.c_int32),
        ("restore_iters", C.c_int32),
        ("mu_init", C.c_double),
        ("kappa_eps", C.c_double),
        ("kappa_mu", C.c_double),
        ("theta_mu", C.c_double),
        ("tau_min", C.c_double),
        ("slack_push", C.c_double),
        ("grad_scale_max", C.c_double),
        ("reach_screen", C.c_int32),
        ("slack_start", C.c_int32),
        ("dual_inf_tol", C.c_double),      # [0.4.0] IPOPT's complete termination test: the three UNSCALED tolerances
        ("constr_viol_tol", C.c_double),
        ("compl_inf_tol", C.c_double),
        ("stall_iters", C.c_int32),        # [0.4.0] the stall rule's budget (a kernel constant until 0.3.x)
        ("qp_method", C.c_int32),          # [0.4.0] 0 = Mehrotra predictor-corrector for the all-linear problems, 1 = IPOPT-style filter line search
    ]


# Options that differ from the defaults for every descriptor built through default_opts() from now on (bench.py's
# --no-reach-screen / --slack-start, A/B tools).  Options travel in the descriptor: libcrx has no process-global switches.
OPTS_OVERRIDE = {}


def default_opts(**kw):
    """IPOPT defaults the reference inherits (control.py:593 passes print options only) + libcrx's own two switches
    (include/crx.h crx_ipm_opts: reach_screen = 1, slack_start = 2)."""
    o = IpmOpts(1e-8, 200, 50, 0.1, 10.0, 0.2, 1.5, 0.99, 1e-2, 100.0, 1, 2, 1.0, 1e-4, 1e-4, 100, 0)
    for k, v in {**OPTS_OVERRIDE, **kw}.items():
        setattr(o, k, v)
    return o


def cbf_class_budgets(N, n_obs_max):
    """(stall_iters, restore_iters) crx_cbf_desc_default picks for a problem class; explicit OPTS_OVERRIDE entries win."""
    kw = {"stall_iters": 50, "restore_iters": 25} if (int(N) <= 12 and int(n_obs_max) <= 1) else {"stall_iters": 100, "restore_iters": 50}
    return {k: v for k, v in kw.items() if k not in OPTS_OVERRIDE}


class PlannerDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("reserved0", C.c_int32),
        ("A", C.c_double * 36),
        ("B", C.c_double * 12),
        ("w_ref", C.c_double),
        ("w_dey", C.c_double),
        ("w_prog", C.c_double),
        ("vx_max", C.c_double),
        ("delta_max", C.c_double),
        ("a_max", C.c_double),
        ("dt_ref", C.c_double),
        ("fallback_gain", C.c_double),
        ("opts", IpmOpts),
    ]


class CbfDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("n_obs_max", C.c_int32),
        ("per_stage_target", C.c_int32),
        ("degree", C.c_int32),
        ("A", C.c_double * 36),
        ("B", C.c_double * 12),
        ("Q", C.c_double * 6),
        ("R", C.c_double * 2),
        ("delta_max", C.c_double),
        ("a_max", C.c_double),
        ("v_min", C.c_double),
        ("v_max", C.c_double),
        ("ey_max", C.c_double),
        ("alpha", C.c_double),
        ("margin", C.c_double),
        ("l_sum", C.c_double),
        ("w_sum", C.c_double),
        ("w_slack", C.c_double),
        ("opts", IpmOpts),
    ]


class LmpcDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("n_ss_max", C.c_int32),
        ("Q", C.c_double * 6),
        ("R", C.c_double * 2),
        ("dR", C.c_double * 2),
        ("x_track", C.c_double * 6),
        ("v_max", C.c_double),
        ("ey_max", C.c_double),
        ("delta_max", C.c_double),
        ("a_max", C.c_double),
        ("w_x0", C.c_double),
        ("opts", IpmOpts),
    ]


class PrepDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("n_veh_max", C.c_int32),
        ("n_opt", C.c_int32),
        ("reserved0", C.c_int32),
        ("prediction_factor", C.c_double),
        ("lookahead", C.c_double),
        ("track_width", C.c_double),
        ("lap_length", C.c_double),
        ("veh_length", C.c_double),
        ("veh_width", C.c_double),
        ("safety_margin", C.c_double),
        ("dt_ref", C.c_double),
    ]


class LmpcPrepDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("n_points", C.c_int32), ("n_laps", C.c_int32), ("n_ss_per_lap", C.c_int32),
        ("n_ss_laps", C.c_int32), ("max_neighbours", C.c_int32), ("n_seg", C.c_int32), ("shift", C.c_int32),
        ("bandwidth", C.c_double), ("scale", C.c_double * 5), ("dt", C.c_double), ("lap_length", C.c_double),
    ]


def lmpcprep_desc(N, n_points, n_laps, n_seg, dt, lap_length, n_ss_per_lap=22, n_ss_laps=2, max_neighbours=40, shift=0):
    """Literals of control/lmpc_helper.py:42-57, utils/base.py:605 and LMPCRacingParam (utils/base.py:351-376)."""
    return LmpcPrepDesc(int(N), int(n_points), int(n_laps), int(n_ss_per_lap), int(n_ss_laps), int(max_neighbours), int(n_seg),
                        int(shift), 5.0, _arr(C.c_double, 5, (0.1, 1.0, 1.0, 1.0, 1.0)), float(dt), float(lap_length))


class SceneDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("n_all_max", C.c_int32), ("n_veh_max", C.c_int32), ("reserved0", C.c_int32),
                ("safety_factor", C.c_double), ("prediction_factor", C.c_double), ("veh_length", C.c_double), ("lap_length", C.c_double)]


def scene_desc(N, n_all_max, n_veh_max, lap_length, safety_factor=4.5, prediction_factor=0.5, veh_length=0.4):
    """RacingGameParam.safety_factor / planning_prediction_factor, CarParam.length (utils/base.py:379-408, :700)."""
    return SceneDesc(int(N), int(n_all_max), int(n_veh_max), 0, safety_factor, prediction_factor, veh_length, float(lap_length))


class PathDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("reserved0", C.c_int32), ("alpha", C.c_double), ("w_rate", C.c_double), ("opts", IpmOpts)]


class PlantDesc(C.Structure):
    _fields_ = [
        ("n_sub", C.c_int32),
        ("n_seg", C.c_int32),
        ("dt_sub", C.c_double),
        ("lap_length", C.c_double),
        ("m", C.c_double), ("lf", C.c_double), ("lr", C.c_double), ("Iz", C.c_double),
        ("Df", C.c_double), ("Cf", C.c_double), ("Bf", C.c_double),
        ("Dr", C.c_double), ("Cr", C.c_double), ("Br", C.c_double),
    ]


class SelectDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("n_veh_max", C.c_int32),
        ("veh_length", C.c_double),
        ("veh_width", C.c_double),
        ("lap_length", C.c_double),
        ("w_prog", C.c_double),
        ("w_coll", C.c_double),
        ("w_switch", C.c_double),
    ]


def _arr(ctype, n, values):
    return (ctype * n)(*[float(v) for v in np.asarray(values, dtype=float).reshape(-1)])


def planner_desc(N, A, B, opts=None):
    """Constants hard-coded in the reference's planner (overtake_traj_planner.py:276-334,367)."""
    return PlannerDesc(
        int(N), 0, _arr(C.c_double, 36, A), _arr(C.c_double, 12, B),
        20.0, 30.0, 200.0, 5.0, 0.5, 1.5, 0.1, 1.1, opts or default_opts(),
    )


def cbf_desc(N, n_obs_max, A, B, Q=(10.0, 0.0, 0.0, 4.0, 0.0, 40.0), R=(0.1, 0.1), alpha=0.8,
             margin=0.2, ey_max=1.0, per_stage_target=False, delta_max=0.5, a_max=1.0, v_min=0.0,
             v_max=10.0, l_sum=0.4, w_sum=0.2, w_slack=1e4, degree=6, opts=None):
    """Defaults = MPCCBFRacingParam / SystemParam / CarParam (utils/base.py:272-291,708-713,699-705)
    and the literals in control.mpccbf (control.py:527-528,560).  Without `opts` the budgets follow the problem class like
    crx_cbf_desc_default (include/crx.h crx_ipm_opts.stall_iters): (50, 25) for N <= 12 with at most one obstacle slot, (100, 50) otherwise."""
    if opts is None:
        opts = default_opts(**cbf_class_budgets(N, n_obs_max))
    return CbfDesc(
        int(N), int(n_obs_max), int(bool(per_stage_target)), int(degree),
        _arr(C.c_double, 36, A), _arr(C.c_double, 12, B), _arr(C.c_double, 6, Q),
        _arr(C.c_double, 2, R), delta_max, a_max, v_min, v_max, ey_max, alpha, margin, l_sum,
        w_sum, w_slack, opts or default_opts(),
    )


def lmpc_desc(N=12, n_ss_max=44, Q=(0.0,) * 6, R=(1.0, 0.25), dR=(4.0, 0.0), x_track=(5.0, 0, 0, 0, 0, 0),
              v_max=10.0, ey_max=1.0, delta_max=0.5, a_max=1.0, w_x0=1e4, opts=None):
    """Defaults = LMPCRacingParam (utils/base.py:350-376), SystemParam (:708-713) and the literal
    x_track of control.lmpc (control.py:649)."""
    return LmpcDesc(int(N), int(n_ss_max), _arr(C.c_double, 6, Q), _arr(C.c_double, 2, R), _arr(C.c_double, 2, dR),
                    _arr(C.c_double, 6, x_track), v_max, ey_max, delta_max, a_max, w_x0, opts or default_opts())


def prep_desc(N, n_veh_max, n_opt, track_width, lap_length, prediction_factor=0.5, veh_length=0.4, veh_width=0.2):
    """planner_helper.py:51-53 and overtake_traj_planner.py:288,296 literals."""
    return PrepDesc(int(N), int(n_veh_max), int(n_opt), 0, prediction_factor, 4.0, track_width, lap_length,
                    veh_length, veh_width, 0.15, 0.1)


def path_desc(N, alpha, w_rate=100.0, opts=None):
    """overtake_path_planner.py:246-253 literals."""
    return PathDesc(int(N), 0, float(alpha), float(w_rate), opts or default_opts())


def plant_desc(n_seg, lap_length, timestep=0.1, dt_sub=0.001):
    """BicycleDynamicsParam defaults (utils/base.py:686-697) and the sub-step loop of base.py:899-905."""
    n_sub = 0
    while (n_sub + 1) * dt_sub <= timestep:
        n_sub += 1
    D = 0.8 * 1.98 * 9.81 / 2.0
    return PlantDesc(n_sub, int(n_seg), dt_sub, lap_length, 1.98, 0.125, 0.125, 0.024, D, 1.25, 1.0, D, 1.25, 1.0)


def select_desc(N, n_veh_max, lap_length, veh_length=0.4, veh_width=0.2):
    """overtake_traj_planner.py:209,223,243 literals."""
    return SelectDesc(int(N), int(n_veh_max), veh_length, veh_width, lap_length, 10.0, 100.0, 100.0)


_D = np.float64
_I = np.int32


def _in(a, dtype, shape):
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.shape != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Binding:
    """Host-pointer entry points of a library exporting <prefix>planner_solve / cbf_solve / select."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        for name in ("planner_solve", "cbf_solve", "select", "lmpc_solve", "planner_prep", "plant_step", "path_solve"):
            if hasattr(lib, prefix + name):
                getattr(lib, prefix + name).restype = C.c_int
        self._check = None

    def _call(self, name, *args):
        rc = getattr(self.lib, self.prefix + name)(*args)
        if rc != 0:
            msg = ""
            if hasattr(self.lib, self.prefix + "last_error"):
                f = getattr(self.lib, self.prefix + "last_error")
                f.restype = C.c_char_p
                msg = (f() or b"").decode()
            raise RuntimeError("%s%s failed: rc=%d %s" % (self.prefix, name, rc, msg))

    def planner_solve(self, desc, x0, bez_s, bez_ey, ey_lb, ey_ub):
        N = desc.N
        x0 = np.ascontiguousarray(x0, dtype=_D)
        Bn = x0.shape[0]
        x0 = _in(x0, _D, (Bn, 6))
        bez_s = _in(bez_s, _D, (Bn, N + 1))
        bez_ey = _in(bez_ey, _D, (Bn, N + 1))
        ey_lb = _in(ey_lb, _D, (Bn, N))
        ey_ub = _in(ey_ub, _D, (Bn,))
        out = dict(
            X=np.zeros((Bn, N + 1, 6)), U=np.zeros((Bn, N, 2)), cost=np.zeros(Bn),
            status=np.zeros(Bn, dtype=_I), kkt=np.zeros(Bn), iters=np.zeros(Bn, dtype=_I),
        )
        self._call(
            "planner_solve", C.byref(desc), C.c_int(Bn), _p(x0), _p(bez_s), _p(bez_ey), _p(ey_lb),
            _p(ey_ub), _p(out["X"]), _p(out["U"]), _p(out["cost"]), _p(out["status"]),
            _p(out["kkt"]), _p(out["iters"]),
        )
        return out

    def cbf_solve(self, desc, x0, xt, obs_s, obs_ey, lap_off, n_obs, obs_dims=None):
        """crx_cbf_solve; with obs_dims (Bn, V, 2) = (l_agent + l_obs, w_agent + w_obs) per obstacle slot: crx_cbf_solve_dims."""
        N, V = desc.N, desc.n_obs_max
        x0 = np.ascontiguousarray(x0, dtype=_D)
        Bn = x0.shape[0]
        x0 = _in(x0, _D, (Bn, 6))
        xt = _in(xt, _D, (Bn, N + 1, 6) if desc.per_stage_target else (Bn, 6))
        obs_s = _in(obs_s, _D, (Bn, V, N + 1))
        obs_ey = _in(obs_ey, _D, (Bn, V, N + 1))
        lap_off = _in(lap_off, _D, (Bn, V))
        n_obs = _in(n_obs, _I, (Bn,))
        out = dict(
            X=np.zeros((Bn, N + 1, 6)), U=np.zeros((Bn, N, 2)), sigma=np.zeros((Bn, V, N + 1)),
            cost=np.zeros(Bn), status=np.zeros(Bn, dtype=_I), kkt=np.zeros(Bn),
            iters=np.zeros(Bn, dtype=_I),
        )
        if obs_dims is not None:
            obs_dims = _in(obs_dims, _D, (Bn, V, 2))
            self._call(
                "cbf_solve_dims", C.byref(desc), C.c_int(Bn), _p(x0), _p(xt), _p(obs_s), _p(obs_ey),
                _p(lap_off), _p(n_obs), _p(obs_dims), _p(out["X"]), _p(out["U"]), _p(out["sigma"]), _p(out["cost"]),
                _p(out["status"]), _p(out["kkt"]), _p(out["iters"]),
            )
            return out
        self._call(
            "cbf_solve", C.byref(desc), C.c_int(Bn), _p(x0), _p(xt), _p(obs_s), _p(obs_ey),
            _p(lap_off), _p(n_obs), _p(out["X"]), _p(out["U"]), _p(out["sigma"]), _p(out["cost"]),
            _p(out["status"]), _p(out["kkt"]), _p(out["iters"]),
        )
        return out

    def select(self, desc, n_veh, X, obs_s, obs_ey, old_flag):
        N, V = desc.N, desc.n_veh_max
        n_veh = np.ascontiguousarray(n_veh, dtype=_I)
        S = n_veh.shape[0]
        X = _in(X, _D, (S, V + 1, N + 1, 6))
        obs_s = _in(obs_s, _D, (S, V, N + 1))
        obs_ey = _in(obs_ey, _D, (S, V, N + 1))
        old_flag = _in(old_flag, _I, (S,))
        out = dict(flag=np.zeros(S, dtype=_I), sel_cost=np.zeros((S, V + 1)), best_X=np.zeros((S, N + 1, 6)))
        self._call(
            "select", C.byref(desc), C.c_int(S), _p(n_veh), _p(X), _p(obs_s), _p(obs_ey),
            _p(old_flag), _p(out["flag"]), _p(out["sel_cost"]), _p(out["best_X"]),
        )
        return out

    def planner_plan(self, desc, sdesc, x0, bez_s, bez_ey, ey_lb, ey_ub, n_veh, obs_s, obs_ey, old_flag):
        """Fused planner step (crx_planner_plan): all region QPs of every scenario + the selection.
        Arrays of the QP part have leading dimension S*(V+1), scenario-major.  Only libcrx exports it;
        the oracle composes planner_solve + select (tests compare the two)."""
        N, V = desc.N, sdesc.n_veh_max
        n_veh = np.ascontiguousarray(n_veh, dtype=_I)
        S = n_veh.shape[0]
        Bn = S * (V + 1)
        x0 = _in(x0, _D, (Bn, 6))
        bez_s = _in(bez_s, _D, (Bn, N + 1))
        bez_ey = _in(bez_ey, _D, (Bn, N + 1))
        ey_lb = _in(ey_lb, _D, (Bn, N))
        ey_ub = _in(ey_ub, _D, (Bn,))
        obs_s = _in(obs_s, _D, (S, V, N + 1))
        obs_ey = _in(obs_ey, _D, (S, V, N + 1))
        old_flag = _in(old_flag, _I, (S,))
        out = dict(
            X=np.zeros((Bn, N + 1, 6)), U=np.zeros((Bn, N, 2)), cost=np.zeros(Bn),
            status=np.zeros(Bn, dtype=_I), kkt=np.zeros(Bn), iters=np.zeros(Bn, dtype=_I),
            flag=np.zeros(S, dtype=_I), sel_cost=np.zeros((S, V + 1)), best_X=np.zeros((S, N + 1, 6)),
        )
        fn = getattr(self.lib, self.prefix + "planner_plan")
        fn.restype = C.c_int
        self._call(
            "planner_plan", C.byref(desc), C.byref(sdesc), C.c_int(S), _p(x0), _p(bez_s), _p(bez_ey),
            _p(ey_lb), _p(ey_ub), _p(n_veh), _p(obs_s), _p(obs_ey), _p(old_flag), _p(out["X"]),
            _p(out["U"]), _p(out["cost"]), _p(out["status"]), _p(out["kkt"]), _p(out["iters"]),
            _p(out["flag"]), _p(out["sel_cost"]), _p(out["best_X"]),
        )
        return out

    def lmpc_solve(self, desc, x0, u_old, A, B, Cm, ss, qfun, n_ss=None):
        """crx_lmpc_solve.  A (Bn,N,6,6), B (Bn,N,6,2), Cm (Bn,N,6), ss (Bn,6,M), qfun (Bn,M)."""
        N, M = desc.N, desc.n_ss_max
        x0 = np.ascontiguousarray(x0, dtype=_D)
        Bn = x0.shape[0]
        x0 = _in(x0, _D, (Bn, 6))
        u_old = _in(u_old, _D, (Bn, 2))
        A = _in(np.asarray(A, dtype=_D).reshape(Bn, N, 36), _D, (Bn, N, 36))
        B = _in(np.asarray(B, dtype=_D).reshape(Bn, N, 12), _D, (Bn, N, 12))
        Cm = _in(np.asarray(Cm, dtype=_D).reshape(Bn, N, 6), _D, (Bn, N, 6))
        ss = _in(ss, _D, (Bn, 6, M))
        qfun = _in(qfun, _D, (Bn, M))
        n_ss = np.full(Bn, M, dtype=_I) if n_ss is None else _in(n_ss, _I, (Bn,))
        out = dict(
            X=np.zeros((Bn, N + 1, 6)), U=np.zeros((Bn, N, 2)), lam=np.zeros((Bn, M)), cost=np.zeros(Bn),
            status=np.zeros(Bn, dtype=_I), kkt=np.zeros(Bn), iters=np.zeros(Bn, dtype=_I),
        )
        self._call(
            "lmpc_solve", C.byref(desc), C.c_int(Bn), _p(x0), _p(u_old), _p(A), _p(B), _p(Cm), _p(ss), _p(qfun),
            _p(n_ss), _p(out["X"]), _p(out["U"]), _p(out["lam"]), _p(out["cost"]), _p(out["status"]),
            _p(out["kkt"]), _p(out["iters"]),
        )
        return out

    def lmpc_prep(self, desc, ss_xcurv, u_ss, qfun, time_ss, it, x, lin_points, lin_input, track, from_plan=False, seed=None):
        """crx_lmpc_prep: stage models + safe-set selection.  ss_xcurv (Bn,L,P,6), u_ss (Bn,L,P,2), qfun (Bn,L,P).
        A, B, C are in/out (a singular stage keeps its three regression rows): `seed` = (A, B, C) to start from, else zeros."""
        N, P, L, M = desc.N, desc.n_points, desc.n_laps, desc.n_ss_per_lap * desc.n_ss_laps
        x = np.ascontiguousarray(x, dtype=_D)
        Bn = x.shape[0]
        ss_xcurv, u_ss, qfun = _in(ss_xcurv, _D, (Bn, L, P, 6)), _in(u_ss, _D, (Bn, L, P, 2)), _in(qfun, _D, (Bn, L, P))
        time_ss, it, x = _in(time_ss, _I, (Bn, L)), _in(it, _I, (Bn,)), _in(x, _D, (Bn, 6))
        lin_points, lin_input = _in(lin_points, _D, (Bn, N + 1, 6)), _in(lin_input, _D, (Bn, N, 2))
        track = _in(track, _D, (desc.n_seg, 6))
        out = dict(A=np.zeros((Bn, N, 6, 6)), B=np.zeros((Bn, N, 6, 2)), C=np.zeros((Bn, N, 6)), ss=np.zeros((Bn, 6, M)),
                   qfun=np.zeros((Bn, M)), status=np.zeros(Bn, dtype=_I))
        if seed is not None:
            out["A"], out["B"], out["C"] = (_in(a, _D, out[k].shape).copy() for k, a in zip("ABC", seed))
        getattr(self.lib, self.prefix + "lmpc_prep").restype = C.c_int
        self._call("lmpc_prep", C.byref(desc), C.c_int(Bn), _p(ss_xcurv), _p(u_ss), _p(qfun), _p(time_ss), _p(it), _p(x),
                   _p(lin_points), _p(lin_input), C.c_int(int(bool(from_plan))), _p(track), _p(out["A"]), _p(out["B"]),
                   _p(out["C"]), _p(out["ss"]), _p(out["qfun"]), _p(out["status"]))
        return out

    def planner_scene(self, desc, ego_xcurv, n_all, veh_xcurv, pred_s, pred_ey):
        """crx_planner_scene: interest test, partial sort, veh_infos, max_delta_v, predictions in sorted order."""
        N1, VA, V = desc.N + 1, desc.n_all_max, desc.n_veh_max
        n_all = np.ascontiguousarray(n_all, dtype=_I)
        S = n_all.shape[0]
        ego_xcurv, veh_xcurv = _in(ego_xcurv, _D, (S, 6)), _in(veh_xcurv, _D, (S, VA, 6))
        pred_s, pred_ey = _in(pred_s, _D, (S, VA, N1)), _in(pred_ey, _D, (S, VA, N1))
        out = dict(n_veh=np.zeros(S, dtype=_I), overflow=np.zeros(S, dtype=_I), order=np.zeros((S, V), dtype=_I), veh_info=np.zeros((S, V, 3)),
                   max_dv=np.zeros(S), obs_s=np.zeros((S, V, N1)), obs_ey=np.zeros((S, V, N1)))
        getattr(self.lib, self.prefix + "planner_scene").restype = C.c_int
        self._call("planner_scene", C.byref(desc), C.c_int(S), _p(ego_xcurv), _p(n_all), _p(veh_xcurv), _p(pred_s), _p(pred_ey),
                   _p(out["n_veh"]), _p(out["overflow"]), _p(out["order"]), _p(out["veh_info"]), _p(out["max_dv"]), _p(out["obs_s"]),
                   _p(out["obs_ey"]))
        return out

    def planner_prep(self, desc, x_wrapped, x_raw, n_veh, veh_info, max_dv, obs_s, obs_ey, opt_s, opt_ey):
        """crx_planner_prep: Bezier references + ey bounds of every region, in crx_planner_solve's layout."""
        N, V, T = desc.N, desc.n_veh_max, desc.n_opt
        n_veh = np.ascontiguousarray(n_veh, dtype=_I)
        S = n_veh.shape[0]
        R = V + 1
        x_wrapped, x_raw = _in(x_wrapped, _D, (S, 6)), _in(x_raw, _D, (S, 6))
        veh_info = _in(veh_info, _D, (S, V, 3))
        max_dv = _in(max_dv, _D, (S,))
        obs_s, obs_ey = _in(obs_s, _D, (S, V, N + 1)), _in(obs_ey, _D, (S, V, N + 1))
        opt_s, opt_ey = _in(opt_s, _D, (T,)), _in(opt_ey, _D, (T,))
        out = dict(x0=np.zeros((S * R, 6)), bez_s=np.zeros((S * R, N + 1)), bez_ey=np.zeros((S * R, N + 1)),
                   ey_lb=np.zeros((S * R, N)), ey_ub=np.zeros(S * R))
        self._call("planner_prep", C.byref(desc), C.c_int(S), _p(x_wrapped), _p(x_raw), _p(n_veh), _p(veh_info),
                   _p(max_dv), _p(obs_s), _p(obs_ey), _p(opt_s), _p(opt_ey), _p(out["x0"]), _p(out["bez_s"]),
                   _p(out["bez_ey"]), _p(out["ey_lb"]), _p(out["ey_ub"]))
        return out

    def plant_step(self, desc, track, xglob, xcurv, u):
        """crx_plant_step: one control step of the zero-noise plant for a batch of vehicles."""
        xglob = np.ascontiguousarray(xglob, dtype=_D)
        Bn = xglob.shape[0]
        track = _in(track, _D, (desc.n_seg, 6))
        xglob, xcurv, u = _in(xglob, _D, (Bn, 6)), _in(xcurv, _D, (Bn, 6)), _in(u, _D, (Bn, 2))
        out = dict(xglob=np.zeros((Bn, 6)), xcurv=np.zeros((Bn, 6)))
        self._call("plant_step", C.byref(desc), C.c_int(Bn), _p(track), _p(xglob), _p(xcurv), _p(u), _p(out["xglob"]),
                   _p(out["xcurv"]))
        return out

    def path_solve(self, desc, opt, bez, lb, ub, e0, eN):
        """crx_path_solve: the 1-D QPs of the overtake path planner, one per candidate region."""
        N = desc.N
        e0 = np.ascontiguousarray(e0, dtype=_D)
        Bn = e0.shape[0]
        opt, bez, lb, ub = (_in(a, _D, (Bn, N + 1)) for a in (opt, bez, lb, ub))
        e0, eN = _in(e0, _D, (Bn,)), _in(eN, _D, (Bn,))
        out = dict(E=np.zeros((Bn, N + 1)), cost=np.zeros(Bn), status=np.zeros(Bn, dtype=_I), kkt=np.zeros(Bn),
                   iters=np.zeros(Bn, dtype=_I))
        self._call("path_solve", C.byref(desc), C.c_int(Bn), _p(opt), _p(bez), _p(lb), _p(ub), _p(e0), _p(eN), _p(out["E"]),
                   _p(out["cost"]), _p(out["status"]), _p(out["kkt"]), _p(out["iters"]))
        return out
