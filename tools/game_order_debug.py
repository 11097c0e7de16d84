"""GPU: does the previous control step's iteration count predict this step's?  Learning-MPC laps (bench.py --workload game), one
batch: per step the solver kernel's own time (crx_set_timing) with index order and with longest-first dispatch, and the rank
correlation of consecutive iteration counts.  Usage: python tools/game_order_debug.py [steps]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "car-racing_amd"))
import bench   # noqa: E402
import crx   # noqa: E402
from crx import torch_api   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
crx.init(0)
L = crx.lib()
L.crx_last_kernel_ms.restype = __import__("ctypes").c_double
out = {}
for mode in ("index", "longest_first"):
    args = argparse.Namespace(race_streams=1, dispatch=mode)
    cx = bench.Ctx()
    w = bench.make_game(cx, args, 4096)
    part = w.step.__self__.parts[0]
    real = torch_api.lmpc_solve_dev
    ms, its = [], []

    def timed(*a, **k):
        L.crx_set_timing(1)
        r = real(*a, **k)
        torch.cuda.synchronize()
        ms.append(L.crx_last_kernel_ms())
        L.crx_set_timing(0)
        return r
    torch_api.lmpc_solve_dev = timed
    for _ in range(steps):
        w.step()
        torch.cuda.synchronize()
        its.append(part.ws.iters.cpu().numpy().copy())
    torch_api.lmpc_solve_dev = real
    out[mode] = (np.array(ms), np.array(its))
for mode, (ms, its) in out.items():
    print(mode, "solver kernel ms per step: first 5", np.round(ms[:5], 3), "mean", ms.mean().round(3), "last 5", np.round(ms[-5:], 3))
its = out["index"][1]
rk = lambda a: np.argsort(np.argsort(a))   # noqa: E731
print("rank correlation of iterations, step k vs k+1:", np.round([np.corrcoef(rk(its[k]), rk(its[k + 1]))[0, 1] for k in range(0, steps - 1, max(1, steps // 12))], 2))
print("iterations per step: mean", its.mean(axis=1).round(1)[:: max(1, steps // 12)], "max", its.max(axis=1)[:: max(1, steps // 12)])
print("same bits in both modes:", bool((out["index"][1] == out["longest_first"][1]).all()))
