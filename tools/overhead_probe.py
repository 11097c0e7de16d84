"""GPU: the fixed cost of a planner QP (set-up, first evaluation, write-back) against its iterations: one launch of 65536 region
QPs (the cfg5 shard) with max_iter = 1, 2, 3, 4, 8 and the default, and the same for the 1-obstacle NLP.  time(max_iter) is close to
linear for small caps: the intercept is the fixed cost, the slope one iteration of every QP still running.
Usage: python tools/overhead_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "car-racing_amd"))
import crx   # noqa: E402
from crx import abi, synth, torch_api   # noqa: E402

crx.init(0)
dev = torch.device("cuda", 0)
A, B = synth.load_AB()
t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
p = synth.cfg3_planner(4096, N=12, seed=3)
args = [t(np.tile(p[k], (4,) + (1,) * (p[k].ndim - 1))) for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")]
n = args[0].shape[0]


def run(f, reps=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for cap in (1, 2, 3, 4, 8, 200):
    o = abi.default_opts(); o.max_iter = cap
    d = abi.planner_desc(12, A, B, opts=o)
    ws = torch_api.PlannerWorkspace(d, n, dev)
    ms = run(lambda: torch_api.planner_solve_dev(d, *args, ws=ws))
    it = ws.iters.cpu().numpy()
    print("planner QPs %d: max_iter %3d  %.3f ms  mean iterations %.2f" % (n, cap, ms, it.mean()))
p = synth.cfg2_mpccbf(16384, N=12, seed=2)
a = [t(p[k]) for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [t(p["n_obs"], torch.int32)]
for cap in (1, 2, 3, 4, 8, 200):
    o = abi.default_opts(); o.max_iter = cap
    d = abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"], opts=o)
    ws = torch_api.CbfWorkspace(d, 16384, dev)
    ms = run(lambda: torch_api.cbf_solve_dev(d, *a, ws=ws))
    print("MPC-CBF NLPs 16384: max_iter %3d  %.3f ms  mean iterations %.2f" % (cap, ms, ws.iters.cpu().numpy().mean()))
