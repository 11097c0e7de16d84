"""GPU: crx_lmpc_prep_dev alone on the state of the learning-MPC laps workload (bench.py --workload game, one batch) after
`steps` control steps.  Usage: [CRX_LIB=...] python tools/prep_probe.py [steps]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "car-racing_amd"))
import bench   # noqa: E402
import crx   # noqa: E402
from crx import torch_api   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
crx.init(0)
cx = bench.Ctx()
w = bench.make_game(cx, argparse.Namespace(race_streams=1, dispatch="index"), 4096)
p = w.step.__self__.parts[0]
for _ in range(steps):
    w.step()
torch.cuda.synchronize()
f = lambda: torch_api.lmpc_prep_dev(p.pdesc, p.ss, p.us, p.qf, p.time_ss, p.it, p.xc, p.ws.X, p.ws.U, p.tab, True, ws=p.pws)   # noqa: E731
f(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    f()
torch.cuda.synchronize()
print("%s: crx_lmpc_prep_dev %.4f ms per 4096 races (after %d steps; time_ss max %d)" % (
    os.environ.get("CRX_LIB", "in-tree"), (time.perf_counter() - t0) / 30 * 1e3, steps, int(p.time_ss.max().item())))
