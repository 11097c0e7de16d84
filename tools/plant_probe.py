import os, sys, time, argparse
import numpy as np, torch
ROOT="/root/repo"
sys.path[:0]=[ROOT, ROOT+"/car-racing_amd"]
import bench, crx
from crx import torch_api
crx.init(0)
cx=bench.Ctx()
w=bench.make_races(cx, argparse.Namespace(race_streams=1, dispatch="index"), 4096)
p=w.step.__self__.parts[0]
for _ in range(10): w.step()
torch.cuda.synchronize()
f=lambda: torch_api.plant_step_wrap_dev(p.pdesc, p.tab, p.xg, p.xc, p.ws.U, 2*p.N, p.xg_next, p.xc_next, p.laps)
f(); torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(50): f()
torch.cuda.synchronize()
print(os.environ.get("CRX_LIB","intree"), "plant %.4f ms per 4096 vehicles"%((time.perf_counter()-t0)/50*1e3), "checksum %.17g"%float(p.xc_next.double().sum().item()))
