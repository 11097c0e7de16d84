"""Lap-time distribution of the batched learning-MPC lap (bench `game` workload: 1024 races from the reference's recorded
safe set, starts perturbed by +-0.03 m/s / +-0.05 m): what tests/test_gpu_closed_loop.py::test_batched_lmpc_laps may assume."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "car-racing_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    import bench
    cx = bench.Ctx()
    w = bench.make_game(cx, argparse.Namespace(), batch=1024)
    laps = w.step.__self__
    s_prev = laps.xc[:, 4].clone()
    done = torch.full((1024,), -1, dtype=torch.int64, device=cx.dev)
    for k in range(300):
        laps.step()
        s = laps.xc[:, 4]
        crossed = (s - s_prev < -5.0) & (done < 0)
        done[crossed] = k + 1
        s_prev = s.clone()
    d = done.cpu().numpy()
    fin = d[d > 0]
    print("finished %d of 1024; lap time steps: min %d p5 %d p50 %d p95 %d max %d" % (len(fin), fin.min(), np.percentile(fin, 5), np.median(fin),
                                                                                        np.percentile(fin, 95), fin.max()))
    print("off track (|ey| > width):", int((laps.xc[:, 5].abs() > 1.0).sum()))


if __name__ == "__main__":
    main()
