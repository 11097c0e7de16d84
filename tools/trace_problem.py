"""Per-iteration phase clocks of ONE problem of the headline batch (needs a -DCRX_PHASE_CLOCKS build: CRX_LIB=tools/ab/libcrx_trace.so).
python tools/trace_problem.py [index, default: the slowest of the batch]"""
import os, sys, ctypes as C
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd"]
import numpy as np
import crx
from crx import abi, synth
gpu = crx.init(); L = crx.lib(); A, B = synth.load_AB()
KEYS = ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")
p = synth.cfg2_mpccbf(256, safe_start=False); d = abi.cbf_desc(12, 1, A, B, alpha=0.8, margin=0.2)
r = gpu.cbf_solve(d, *[p[k] for k in KEYS])
idx = int(sys.argv[1]) if len(sys.argv) > 1 else int(np.argmax(r["iters"]))
names = ["accept/first-order", "adjoint", "mu", "assemble", "riccati back", "forward", "row steps", "line search"]
L.crx_trace_enable(0, 64)
rr = gpu.cbf_solve(d, *[p[k][idx:idx + 1] for k in KEYS])
buf = np.zeros((64, 16)); L.crx_trace_read(buf.ctypes.data_as(C.c_void_p), 64)
n = int(rr["iters"][0])
print("problem %d: iters %d status %d" % (idx, n, int(rr["status"][0])))
print("  it   total | " + " ".join("%9s" % s[:9] for s in names) + " |  mu      alpha     dw")
for i in range(min(n, 64)):
    t = buf[i]
    print("  %2d %7.0f | " % (i, t[8:16].sum()) + " ".join("%9.0f" % v for v in t[8:16]) + " | %.1e %.2e %.1e" % (t[3], t[4], t[6]))
print("  sum of the iterations' clocks: %.0f" % buf[:n, 8:16].sum())
L.crx_trace_enable(0, 0)
