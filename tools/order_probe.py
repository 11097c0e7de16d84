"""GPU: cfg4 launch time under different dispatch orders -- index, longest first by the previous solve's iterations, and an
a-priori key that needs no previous solve (the smallest barrier value h of the start state: a car that starts inside a safety
ellipse is the NLP that needs the restoration phase).  Usage: python tools/order_probe.py [batch]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
A, B = synth.load_AB()
p = synth.cfg4_tracking_cbf(n, N=20, seed=4, safe_start=False)
d = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
a = [t(p[k]) for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [t(p["n_obs"], torch.int32)]
ws = torch_api.CbfWorkspace(d, n, dev)
torch_api.cbf_solve_dev(d, *a, ws=ws)
torch.cuda.synchronize()
it = ws.iters.clone()


def prior_key():
    x0, obs_s, obs_e, lap_off, n_obs = a[0], a[2], a[3], a[4], a[5]
    ds = obs_s[:, :, 0] + lap_off - x0[:, 4:5]
    de = obs_e[:, :, 0] - x0[:, 5:6]
    h = (ds / d.l_sum) ** d.degree + (de / d.w_sum) ** d.degree - 1.0 - d.margin
    h = torch.where(torch.arange(h.shape[1], device=dev)[None, :] < n_obs[:, None], h, torch.full_like(h, 1e30))
    return h.min(dim=1).values


def run(order, reps=15):
    torch_api.cbf_solve_dev(d, *a, ws=ws, order=order)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        torch_api.cbf_solve_dev(d, *a, ws=ws, order=order)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


key = prior_key()
noisy = (it.double() * torch.from_numpy(np.random.default_rng(0).uniform(0.5, 1.5, n)).to(dev))
orders = {"index": None, "longest_first(previous iters)": torch_api.longest_first(it),
          "longest_first(previous iters x U(0.5,1.5))": torch.argsort(noisy, descending=True).to(torch.int32),
          "smallest start barrier first": torch.argsort(key, stable=True).to(torch.int32),
          "start inside an ellipse first (2 classes)": torch.argsort((key >= 0).to(torch.int32), stable=True).to(torch.int32),
          "shortest first (worst case)": torch.argsort(it, stable=True).to(torch.int32),
          "random permutation": torch.from_numpy(np.random.default_rng(1).permutation(n).astype(np.int32)).to(dev)}
for name, o in orders.items():
    print("%-46s %.3f ms per launch of %d" % (name, run(o), n))
orders = {"crx_order_longest_first_dev": lambda: torch_api.longest_first(it), "crx_cbf_order_dev": lambda: torch_api.cbf_order_dev(d, *a),
          "torch: key + argsort": lambda: torch.argsort(prior_key(), stable=True).to(torch.int32)}
for name, f in orders.items():
    o = f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    print("%-46s %.3f ms for the order, %.3f ms per launch with it" % (name, (time.perf_counter() - t0) / 50 * 1e3, run(o)))
itc = it.cpu().numpy(); kc = key.cpu().numpy()
print("iters: inside ellipse %d problems mean %.1f; outside %d mean %.1f" % ((kc < 0).sum(), itc[kc < 0].mean(), (kc >= 0).sum(), itc[kc >= 0].mean()))
