"""GPU: does the racing-game step time depend on how many torch streams the process handed out BEFORE the workload's own six
(bench.py's default run builds the workload after races / game; a stand-alone run builds it first)?
Usage: python tools/stream_offset_probe.py <streams handed out before> [workload]"""
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

pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
wl = sys.argv[2] if len(sys.argv) > 2 else "overtake"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
overlap = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
crx.init(0)
cx = bench.Ctx()
keep = []
for _ in range(pre):
    s = torch.cuda.Stream(device=cx.dev)
    with torch.cuda.stream(s):
        keep.append(torch.zeros(16, device=cx.dev) + 1)
torch.cuda.synchronize()
args = argparse.Namespace(race_streams=K, dispatch="index")
w = {"overtake": bench.make_overtake, "game": bench.make_game, "races": bench.make_races}[wl](cx, args, 4096)
if len(sys.argv) > 4:
    for p_ in w.step.__self__.parts:
        if hasattr(p_, "overlap"):
            p_.overlap = overlap
for _ in range(5):
    w.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    w.step()
torch.cuda.synchronize()
parts = w.step.__self__.parts
ids = [hex(s.cuda_stream & 0xffffff) for s in w.step.__self__.streams]
if len(sys.argv) > 3:   # "torch": the round-2/3 formulation, streams from torch's pool (set up before the first step)
    pass
conc = w.step.__self__
print("%s K=%d, %2d torch streams used before: %.4f ms/step   (%d streams measured to overlap; branch streams %s)" % (
    wl, K, pre, (time.perf_counter() - t0) / 60 * 1e3, conc.n_concurrent, [bool(getattr(p_, "overlap", False)) for p_ in conc.parts]))
