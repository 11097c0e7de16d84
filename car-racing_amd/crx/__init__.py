"""crx -- Python binding of libcrx, the MI355X-native batched OCP solver (include/crx.h).

The library is loaded lazily through a module-level singleton so that objects which use it stay
picklable (the reference pickles its simulator, tests/auto_mpccbf_test.py:42-43) and so that importing
this package never needs a GPU.  There is NO CPU fallback: if libcrx.so is missing or no HIP device
is visible, the first solve raises.
"""
import ctypes
import os

from . import abi, hostprep  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRX_LIB: another build of the same library (A/B measurements of kernel variants, tools/ab_*.sh); default: the in-tree build
LIB_PATH = os.environ.get("CRX_LIB") or os.path.join(_HERE, "libcrx.so")
_state = {"lib": None, "binding": None, "device": None}


class CrxUnavailable(RuntimeError):
    pass


def lib():
    """The raw CDLL (loaded once).  Raises CrxUnavailable if libcrx.so has not been built."""
    if _state["lib"] is None:
        if not os.path.exists(LIB_PATH):
            raise CrxUnavailable(
                "libcrx.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C car-racing_amd/csrc`" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 under the same SONAMEs as /opt/rocm's.
        # Whichever is loaded first serves both libcrx and torch: if libcrx pulled in the system runtime first, a later
        # `import torch` would find "No HIP GPUs" (two HSA runtimes cannot both own the device).  So when torch is
        # installed, let it load its runtime first; libcrx then binds to the same one.  (torch is plumbing for device
        # memory and torch.distributed; the C ABI itself needs none of it.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _state["lib"] = ctypes.CDLL(LIB_PATH)
        _state["lib"].crx_last_error.restype = ctypes.c_char_p
        _state["lib"].crx_timer_ms.restype = ctypes.c_double
    return _state["lib"]


def init(device=None):
    """Bind the library to a HIP device (default: LOCAL_RANK or 0).  Raises without a GPU."""
    L = lib()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _state["device"] != device:
        rc = L.crx_init(int(device))
        if rc != 0:
            raise CrxUnavailable("crx_init(%d) failed: rc=%d %s" % (device, rc, (L.crx_last_error() or b"").decode()))
        _state["device"] = device
        _state["binding"] = abi.Binding(L, "crx_")
    return _state["binding"]


def binding():
    """abi.Binding over the initialised library (initialises on device LOCAL_RANK/0 on first use)."""
    return _state["binding"] or init()


def planner_solve(desc, x0, bez_s, bez_ey, ey_lb, ey_ub):
    return binding().planner_solve(desc, x0, bez_s, bez_ey, ey_lb, ey_ub)


def cbf_solve(desc, x0, xt, obs_s, obs_ey, lap_off, n_obs, obs_dims=None):
    return binding().cbf_solve(desc, x0, xt, obs_s, obs_ey, lap_off, n_obs, obs_dims)


def select(desc, n_veh, X, obs_s, obs_ey, old_flag):
    return binding().select(desc, n_veh, X, obs_s, obs_ey, old_flag)


def planner_plan(desc, sdesc, x0, bez_s, bez_ey, ey_lb, ey_ub, n_veh, obs_s, obs_ey, old_flag):
    return binding().planner_plan(desc, sdesc, x0, bez_s, bez_ey, ey_lb, ey_ub, n_veh, obs_s, obs_ey, old_flag)


def lmpc_solve(desc, x0, u_old, A, B, Cm, ss, qfun, n_ss=None):
    return binding().lmpc_solve(desc, x0, u_old, A, B, Cm, ss, qfun, n_ss)


def path_solve(desc, opt, bez, lb, ub, e0, eN):
    return binding().path_solve(desc, opt, bez, lb, ub, e0, eN)


def plant_step(desc, track, xglob, xcurv, u):
    return binding().plant_step(desc, track, xglob, xcurv, u)
