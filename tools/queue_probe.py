"""GPU: is one of the hardware queues slower than the others?  n streams from crx_streams_create, used for the first time in
creation order; then per stream: (a) 400 back-to-back tiny kernels, (b) 40 launches of a chip-filling solver batch; and (c) the
same solver batch on stream i while stream 0 runs one too.  Usage: GPU_MAX_HW_QUEUES=8 python tools/queue_probe.py [n]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "car-racing_amd"))
import crx   # noqa: E402
from crx import abi, synth, torch_api   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
crx.init(0)
dev = torch.device("cuda", 0)
A, B = synth.load_AB()
p = synth.cfg2_mpccbf(1024, N=12, seed=2, safe_start=True)
d = abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"])
t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
a = [t(p[k]) for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [t(p["n_obs"], torch.int32)]
streams, n_conc = torch_api.new_streams(n, dev)
print("crx_streams_create(%d): the first %d overlap pairwise" % (n, n_conc))
wss = [torch_api.CbfWorkspace(d, 1024, dev) for _ in range(n)]
x = [torch.zeros(64, device=dev) for _ in range(n)]
for i, s in enumerate(streams):       # first use, in creation order
    with torch.cuda.stream(s):
        x[i] += 1
torch.cuda.synchronize()
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        for _ in range(20):
            x[i] += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            x[i] += 1
        torch.cuda.synchronize()
        ta = (time.perf_counter() - t0) / 400 * 1e6
        torch_api.cbf_solve_dev(d, *a, ws=wss[i]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            torch_api.cbf_solve_dev(d, *a, ws=wss[i])
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / 40 * 1e3
    # (c) together with stream 0
    t0 = time.perf_counter()
    for _ in range(40):
        with torch.cuda.stream(streams[0]):
            torch_api.cbf_solve_dev(d, *a, ws=wss[0])
        if i:
            with torch.cuda.stream(s):
                torch_api.cbf_solve_dev(d, *a, ws=wss[i])
    torch.cuda.synchronize()
    tc = (time.perf_counter() - t0) / 40 * 1e3
    print("stream %2d: tiny kernel %.2f us each; 1024 NLPs alone %.4f ms; with stream 0 busy too %.4f ms per pair" % (i, ta, tb, tc))
