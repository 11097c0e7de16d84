"""The slowest problems of the headline batch, each solved ALONE (host-call time of a one-problem launch), and without the crash path: what a
straggler costs outside the 256-wave launch.  python tools/lone_problems.py"""
import sys, os, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "car-racing_amd")]
import numpy as np, torch
import crx
from crx import synth, abi, torch_api
gpu = crx.init(); A, B = synth.load_AB()
KEYS = ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")
p = synth.cfg2_mpccbf(256, safe_start=False)
d = abi.cbf_desc(12, 1, A, B, alpha=0.8, margin=0.2)
r = gpu.cbf_solve(d, *[p[k] for k in KEYS])
it = np.asarray(r["iters"]); order = np.argsort(-it)
def timed(idx, reps=20):
    args = [p[k][idx] for k in KEYS]
    for _ in range(3): gpu.cbf_solve(d, *args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): gpu.cbf_solve(d, *args)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("whole batch: %.3f ms" % timed(np.arange(256)))
for i in list(order[:6]) + list(order[120:123]):
    ms = timed(np.array([i]))
    print("problem %3d iters %2d status %d : lone host-call %.3f ms" % (i, it[i], r["status"][i], ms))
# crash-free reference: the same lone problems with slack_start = 0 (no crash search)
d0 = abi.cbf_desc(12, 1, A, B, alpha=0.8, margin=0.2); d0.opts.slack_start = 0
r0 = gpu.cbf_solve(d0, *[p[k] for k in KEYS]); it0 = np.asarray(r0["iters"])
for i in order[:3]:
    args = [p[k][np.array([i])] for k in KEYS]
    for _ in range(3): gpu.cbf_solve(d0, *args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): gpu.cbf_solve(d0, *args)
    torch.cuda.synchronize(); print("problem %3d slack_start 0: iters %d status %d lone %.3f ms" % (i, it0[i], r0["status"][i], (time.perf_counter() - t0) / 20 * 1e3))
