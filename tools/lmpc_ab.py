"""A/B of two builds of the learning-MPC kernel, bit for bit: the 47 recorded QPs, the noise-floor QPs, and 4096 closed-loop
learning-MPC laps for 40 steps (every state and plan of every race).  Usage: [CRX_LIB=...] python tools/lmpc_ab.py TAG   (dumps
gpurun_out/lmpc_ab_TAG.npz);  python tools/lmpc_ab.py --compare TAG1 TAG2"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd", ROOT + "/tests"]
OUT = ROOT + "/gpurun_out/lmpc_ab_%s.npz"

if sys.argv[1] == "--compare":
    a, b = np.load(OUT % sys.argv[2]), np.load(OUT % sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
    print("compared %d arrays of %s and %s: %s" % (len(a.files), sys.argv[2], sys.argv[3], "IDENTICAL" if not bad else "DIFFERENT in %s" % bad))
    sys.exit(1 if bad else 0)

import torch   # noqa: E402
import bench   # noqa: E402
import crx   # noqa: E402
import helpers   # noqa: E402
from crx import abi   # noqa: E402

gpu = crx.init()
out = {}
g = np.load(ROOT + "/tests/golden/racing_game.npz")
d, args = helpers.lmpc_inputs(g)
r = gpu.lmpc_solve(d, *args)
for k, v in r.items():
    out["rec/" + k] = np.asarray(v)
z = np.load(ROOT + "/tests/golden/lmpc_noise_floor.npz")
r = gpu.lmpc_solve(abi.lmpc_desc(12, 44), *[z[k] for k in ("x0", "u_old", "A", "B", "C", "ss", "qfun", "n_ss")])
for k, v in r.items():
    out["noise/" + k] = np.asarray(v)
cx = bench.Ctx()
w = bench.make_game(cx, argparse.Namespace(race_streams=1, dispatch="index", lap_phase=0), 4096)
p = w.step.__self__.parts[0]
its = []
for _ in range(40):
    w.step()
    torch.cuda.synchronize()      # the step runs on the workload's own stream
    its.append(p.ws.iters.clone())
torch.cuda.synchronize()
out["loop/xc"] = p.xc.cpu().numpy(); out["loop/X"] = p.ws.X.cpu().numpy(); out["loop/U"] = p.ws.U.cpu().numpy()
out["loop/status"] = p.ws.status.cpu().numpy(); out["loop/iters"] = torch.stack(its).cpu().numpy()
os.makedirs(ROOT + "/gpurun_out", exist_ok=True)
np.savez(OUT % sys.argv[1], **out)
print(sys.argv[1], "recorded status", out["rec/status"].tolist()[:12], "loop iterations total", int(out["loop/iters"].sum()))
