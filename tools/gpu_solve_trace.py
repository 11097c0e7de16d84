"""(phase clocks need a library built with -DCRX_PHASE_CLOCKS -- `make TRACE=1`, or tools/build_variant.sh trace "-DCRX_PHASE_CLOCKS" crx_kernels.hip
crx_kernels_obs.hip crx_lmpc.hip + CRX_LIB=tools/ab/libcrx_trace.so, which is what `tools/gpu_pass.sh TAG trace` uses; the default build compiles them out)
Diagnostics: per-phase shader cycles of one crx_solve_kernel problem (hidden crx_trace_* entry points).
usage: gpu_solve_trace.py [cfg2|cfg3|cfg4]"""
import os, sys, ctypes as C
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/car-racing_amd")
import numpy as np
import crx
from crx import abi, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
gpu = crx.init(); L = crx.lib()
A, B = synth.load_AB()
if wl == "cfg3":
    p = synth.cfg3_planner(8, N=12); d = abi.planner_desc(12, A, B)
    rr = gpu.planner_solve(d, p["x0"], p["bez_s"], p["bez_ey"], p["ey_lb"], p["ey_ub"])
    i0 = int(np.argmax(np.asarray(rr["status"]).reshape(len(p["x0"]), -1)[:, 0] == 0))   # a QP that is solved (41 % are screened out before the iteration)
    call = lambda: gpu.planner_solve(d, p["x0"][i0:i0 + 1], p["bez_s"][i0:i0 + 1], p["bez_ey"][i0:i0 + 1], p["ey_lb"][i0:i0 + 1], p["ey_ub"][i0:i0 + 1])
elif wl == "cfg4":
    p = synth.cfg4_tracking_cbf(8, N=20); d = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    call = lambda: gpu.cbf_solve(d, *[p[k][:1] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")])
else:
    p = synth.cfg2_mpccbf(8, N=12); d = abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"])
    call = lambda: gpu.cbf_solve(d, *[p[k][:1] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")])
for mode, names in ((64, ["accept/first-order", "adjoint", "mu", "assemble", "riccati back", "forward", "row steps", "line search"]),
                    (-64, ["ric: T", "ric: H", "ric: set-up", "ric: update", "riccati back", "forward", "row steps", "line search"])):
    L.crx_trace_enable(0, mode)
    r = call()
    buf = np.zeros((64, 16)); L.crx_trace_read(buf.ctypes.data_as(C.c_void_p), 64)
    n = int(r["iters"][0]); tr = buf[1:n]          # skip the first iteration (cold)
    print(wl, "iters", n, "status", int(r["status"][0]))
    for q, nm in enumerate(names):
        print("  %-20s %9.0f" % (nm, tr[:, 8 + q].mean()))
    if mode > 0:
        print("  %-20s %9.0f" % ("TOTAL", tr[:, 8:16].sum(axis=1).mean()))
L.crx_trace_enable(0, 0)
