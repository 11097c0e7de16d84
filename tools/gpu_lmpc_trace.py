# phase clocks need a library built with `make -C car-racing_amd/csrc clean all TRACE=1` (compiled out by default)
"""Diagnostics: per-phase shader cycles of one learning-MPC solve (hidden crx_trace_* entry points)."""
import os, sys, ctypes as C, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/car-racing_amd"); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import crx, helpers
g = np.load(ROOT + "/tests/golden/racing_game.npz")
gpu = crx.init(); L = crx.lib()
d, args = helpers.lmpc_inputs(g)
one = [a[:1] for a in args]
L.crx_trace_enable(0, 64)
r = gpu.lmpc_solve(d, *one)
buf = np.zeros((64, 16)); L.crx_trace_read(buf.ctypes.data_as(C.c_void_p), 64)
n = int(r["iters"][0]); tr = buf[:n]
names = ["rows/grad/e", "lagr+err", "mu", "sigma+lagr", "K asm", "chol K", "W,Lw,T + G scans", "-", "-", "backsub", "row steps", "line search", "accept", "TOTAL"]
print("iters", n, "status", r["status"][0])
for q, nm in enumerate(names):
    print("%-12s %9.0f" % (nm, tr[:, q].mean()))
L.crx_trace_enable(0, 0)
big = [np.concatenate([a] * 88)[:4096] for a in args]
gpu.lmpc_solve(d, *big)
t0 = time.time(); gpu.lmpc_solve(d, *big); print("batch 4096 host-call %.2f ms" % ((time.time() - t0) * 1e3))
