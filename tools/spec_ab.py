"""A/B of the two-wave (speculating) instantiation against the one-wave kernel on the headline batch (GPU box):
which problems differ (there should be none), and what the slowest problems cost ALONE under either kernel.  python tools/spec_ab.py"""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "car-racing_amd")]
import ctypes as C
import numpy as np, torch
import crx
from crx import synth, abi
gpu = crx.init(); L = crx.lib(); A, B = synth.load_AB()
KEYS = ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")
p = synth.cfg2_mpccbf(256, safe_start=False)
d = abi.cbf_desc(12, 1, A, B, alpha=0.8, margin=0.2)
args = [p[k] for k in KEYS]
res = {}
for mode in (0, 2, 1):
    L.crx_debug_speculation(mode)
    res[mode] = gpu.cbf_solve(d, *args)
it = res[0]["iters"]
for mode in (2, 1):
    dx = np.abs(res[mode]["X"] - res[0]["X"]).reshape(256, -1).max(axis=1)
    di = res[mode]["iters"] != it
    bad = np.nonzero((dx > 0) | di)[0]
    print("mode %d vs one-wave kernel: %d problems differ; iters differ on %d; max |dX| %.2e" % (mode, len(bad), di.sum(), dx.max()))
    for b in bad[:12]:
        print("   #%d iters %d -> %d  |dX| %.2e  status %d -> %d" % (b, it[b], res[mode]["iters"][b], dx[b], res[0]["status"][b], res[mode]["status"][b]))
# per-iteration trace of the first differing problem under both kernels (dw column shows the inertia corrections)
def trace(mode, b):
    L.crx_debug_speculation(mode)
    L.crx_trace_enable(0, 64)
    a1 = [p[k][b:b + 1] for k in KEYS]
    r = gpu.cbf_solve(d, *a1)
    buf = np.zeros((64, 16)); L.crx_trace_read(buf.ctypes.data_as(C.c_void_p), 64); L.crx_trace_enable(0, 0)
    return r, buf
order = np.argsort(-it)
def timed(idx, reps=30):
    a1 = [p[k][idx] for k in KEYS]
    for _ in range(3): gpu.cbf_solve(d, *a1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): gpu.cbf_solve(d, *a1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for mode in (0, 2, 1):
    L.crx_debug_speculation(mode)
    print("mode %d: whole batch %.4f ms" % (mode, timed(np.arange(256))))
    for i in order[:8]:
        print("   problem %3d iters %2d : lone host-call %.4f ms" % (i, it[i], timed(np.array([i]))))
for b in order[:3]:
    r0, t0 = trace(0, b)
    print("problem %d: per-iteration dw (one-wave kernel): %s" % (b, " ".join("%.0e" % t0[k][6] for k in range(int(r0["iters"][0])))))
L.crx_debug_speculation(-1)
