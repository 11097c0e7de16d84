"""Iteration-by-iteration comparison of ONE problem between libcrx (hidden crx_trace_* hook: e_d, e_p, e_c, mu, alpha,
alpha_dual, delta_w, accept type per iteration) and the oracle (crx_oracle_set_verbose(1), stderr).  GPU box.

    python tools/parity_trace.py cfg3 12 133        # workload, horizon, problem index in the synthetic batch
"""
import ctypes as C
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (ROOT, os.path.join(ROOT, "car-racing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import crx  # noqa: E402
import oracle  # noqa: E402
from crx import abi, synth  # noqa: E402

wl, N, idx = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-8
gpu, orc, L = crx.init(), oracle.load(), crx.lib()
A, B = synth.load_AB()
if wl == "cfg3":
    p = synth.cfg3_planner(128, N=N)
    d = abi.planner_desc(N, A, B)
    a = tuple(p[k][idx:idx + 1] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub"))
    fg, fo = gpu.planner_solve, orc.planner_solve
elif wl.startswith("cfg4"):      # cfg4:SEED:BATCH -- a problem of the three-obstacle tracking draw (tools/stress_cbf.py, the stress test), e.g. cfg4:11:8192 20 5964
    _, seed, nb = (wl.split(":") + ["4", "256"])[:3]
    p = synth.cfg4_tracking_cbf(int(nb), N=N, seed=int(seed), safe_start=False)
    d = abi.cbf_desc(N, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    a = tuple(p[k][idx:idx + 1] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs"))
    fg, fo = gpu.cbf_solve, orc.cbf_solve
else:
    p = synth.cfg2_mpccbf(256, N=N, safe_start=(wl == "cfg2"))
    d = abi.cbf_desc(N, 1, A, B, alpha=p["alpha"], margin=p["margin"])
    a = tuple(p[k][idx:idx + 1] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs"))
    fg, fo = gpu.cbf_solve, orc.cbf_solve
d.opts.tol = tol
ROWS = 256
L.crx_trace_enable(0, ROWS)
r = fg(d, *a)
buf = np.zeros((ROWS, 16))
L.crx_trace_read(buf.ctypes.data_as(C.c_void_p), ROWS)
L.crx_trace_enable(0, 0)
print("GPU status %d iters %d kkt %.3e cost %.9g" % (r["status"][0], r["iters"][0], r["kkt"][0], r["cost"][0]))
for it in range(min(int(r["iters"][0]) + 1, ROWS)):
    t = buf[it]
    print("gpu it %3d ed %.6e ep %.6e ec %.6e mu %.1e al %.6e a_d %.6e dw %.1e acc %d" % (it, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]))
sys.stdout.flush()
oracle.set_threads(1)
orc.lib.crx_oracle_set_verbose(1)
ro = fo(d, *a)
orc.lib.crx_oracle_set_verbose(0)
print("CPU status %d iters %d kkt %.3e cost %.9g" % (ro["status"][0], ro["iters"][0], ro["kkt"][0], ro["cost"][0]))
