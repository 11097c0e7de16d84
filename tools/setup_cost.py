"""Diagnostics: time of crx_solve_kernel with max_iter = 1, 2, 3 (set-up + write-back vs one iteration), batch 256."""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/car-racing_amd")
import numpy as np, torch
import crx
from crx import abi, synth, torch_api
crx.init(0); L = crx.lib()
import ctypes as C
L.crx_last_kernel_ms.restype = C.c_double
L.crx_set_timing(1)
A, B = synth.load_AB()
dev = torch.device("cuda", 0)
p = synth.cfg2_mpccbf(256, N=12, seed=2)
t_in = [torch.from_numpy(np.ascontiguousarray(p[k])).to(dev) for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [torch.from_numpy(p["n_obs"]).to(dev).to(torch.int32)]
for mi in (1, 2, 3, 5, 3000):
    d = abi.cbf_desc(12, 1, A, B, alpha=p["alpha"], margin=p["margin"])
    d.opts.max_iter = mi
    ws = torch_api.CbfWorkspace(d, 256, dev)
    ts = []
    for _ in range(20):
        torch_api.cbf_solve_dev(d, *t_in, ws=ws); torch.cuda.synchronize(); ts.append(L.crx_last_kernel_ms())
    print("max_iter %4d: kernel %.1f us (iters max %d)" % (mi, 1e3 * np.median(ts[5:]), int(ws.iters.max())))
