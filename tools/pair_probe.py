"""GPU: the closed-loop races on two sub-batch streams, for different PAIRS of streams that all sit on different hardware queues
(crx_streams_create) and for torch's pool streams: is there more to a good pair than "different queues"?
Usage: GPU_MAX_HW_QUEUES=8 python tools/pair_probe.py [workload]"""
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
from crx import montecarlo, torch_api   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "races"
crx.init(0)
cx = bench.Ctx()
cand, nc = torch_api.new_streams(8, cx.dev)
print("crx_streams_create(8): %d overlap pairwise" % nc)
pool = [torch.cuda.Stream(device=cx.dev) for _ in range(8)]
args = argparse.Namespace(race_streams=2, dispatch="index")
make = {"overtake": bench.make_overtake, "game": bench.make_game, "races": bench.make_races}[wl]


def run(streams, tag):
    w = make(cx, args, 4096)
    conc = w.step.__self__
    if streams is not None:
        conc2 = montecarlo.Concurrent(conc.parts, cx.dev, streams=streams)
        for p in conc2.parts:
            if hasattr(p, "overlap"):
                p.overlap = False
        step = conc2.step
    else:
        step = conc.step
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        step()
    torch.cuda.synchronize()
    print("%s %-34s %.4f ms/step" % (wl, tag, (time.perf_counter() - t0) / 60 * 1e3))


run(None, "Concurrent's own choice")
for j in range(1, nc):
    run([cand[0], cand[j]], "libcrx streams 0 and %d" % j)
run([cand[1], cand[2]], "libcrx streams 1 and 2")
for j in (1, 2, 3, 4):
    run([pool[0], pool[j]], "torch pool streams 0 and %d" % j)
run([torch.cuda.current_stream(cx.dev), cand[0]], "the default stream and libcrx 0")
