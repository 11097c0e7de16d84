"""A/B of two builds of the solver kernel, bit for bit: the cfg2 draw (4096 NLPs), the cfg4 draw (4096), planner QPs (N = 12 and 10), 30
steps of the racing game with traffic (1024 races) and 40 steps of the MPC-CBF races (4096).  Usage: [CRX_LIB=...] python tools/cbf_ab.py TAG; python tools/cbf_ab.py --compare A B"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd", ROOT + "/tests"]
OUT = ROOT + "/gpurun_out/cbf_ab_%s.npz"
if sys.argv[1] == "--compare":
    a, b = np.load(OUT % sys.argv[2]), np.load(OUT % sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
    print("compared %d arrays of %s and %s: %s" % (len(a.files), sys.argv[2], sys.argv[3], "IDENTICAL" if not bad else "DIFFERENT in %s" % bad))
    sys.exit(1 if bad else 0)
import torch   # noqa: E402
import bench   # noqa: E402
import crx   # noqa: E402
from crx import abi, synth   # noqa: E402

gpu = crx.init()
A, B = synth.load_AB()
out = {}
p = synth.cfg2_mpccbf(4096, N=12, seed=2, safe_start=False)
r = gpu.cbf_solve(abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"]), *[p[k] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")])
for k, v in r.items():
    out["cfg2/" + k] = np.asarray(v)
p = synth.cfg4_tracking_cbf(4096, N=20, seed=4, safe_start=False)
r = gpu.cbf_solve(abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True),
                  *[p[k] for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")])
for k, v in r.items():
    out["cfg4/" + k] = np.asarray(v)
p = synth.cfg3_planner(1024, N=12, seed=3)
r = gpu.planner_solve(abi.planner_desc(12, A, B), *[p[k] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")])
for k, v in r.items():
    out["cfg3/" + k] = np.asarray(v)
p = synth.cfg3_planner(256, N=10, seed=5)
r = gpu.planner_solve(abi.planner_desc(10, A, B), *[p[k] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")])
for k, v in r.items():
    out["plan10/" + k] = np.asarray(v)
cx = bench.Ctx()
wo = bench.make_overtake(cx, argparse.Namespace(race_streams=1, dispatch="index", lap_phase=0), 1024)
po = wo.step.__self__.parts[0]
for _ in range(30):
    wo.step()
torch.cuda.synchronize()
out["overtake/xc"] = po.lm.xc.cpu().numpy(); out["overtake/tU"] = po.tws.U.cpu().numpy(); out["overtake/titers"] = po.tws.iters.cpu().numpy()
w = bench.make_races(cx, argparse.Namespace(race_streams=1, dispatch="index", lap_phase=0), 4096)
pr = w.step.__self__.parts[0]
for _ in range(40):
    w.step()
torch.cuda.synchronize()
out["races/xc"] = pr.xc.cpu().numpy(); out["races/U"] = pr.ws.U.cpu().numpy(); out["races/iters"] = pr.ws.iters.cpu().numpy()
np.savez(OUT % sys.argv[1], **out)
print(sys.argv[1], "cfg2 iterations", int(out["cfg2/iters"].sum()), "cfg4", int(out["cfg4/iters"].sum()), "races", int(out["races/iters"].sum()))
