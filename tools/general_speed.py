"""Speed of the GENERAL instantiations (csrc/crx_kernels_gen.hip: every horizon other than 10 / 12 / 20) next to the tuned ones on the same draw:
    [CRX_LIB=tools/ab/libcrx_NAME.so] python tools/general_speed.py
cfg2-style NLPs (1 obstacle) at N = 11 / 12 / 13 and cfg4-style (3 obstacles) at N = 19 / 20 / 21, batch 4096, device-resident, HIP-event time."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd"]
import numpy as np
import torch
import crx
from crx import abi, synth, torch_api
crx.init(0); A, B = synth.load_AB(); dev = torch.device("cuda", 0)
t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)   # noqa: E731
for kind, Ns in (("cfg2", (11, 12, 13, 15, 16, 17, 20)), ("cfg4", (15, 16, 17, 19, 20, 21))):
    for N in Ns:
        Bn = 4096
        if kind == "cfg2":
            p = synth.cfg2_mpccbf(Bn, N=N, seed=2, safe_start=False); d = abi.cbf_desc(N, 1, A, B, alpha=p["alpha"], margin=p["margin"])
        else:
            p = synth.cfg4_tracking_cbf(Bn, N=N, seed=4, safe_start=False); d = abi.cbf_desc(N, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
        a = [t(p[k]) for k in ("x0", "xt", "obs_s", "obs_ey", "lap_off")] + [t(p["n_obs"], torch.int32)]
        ws = torch_api.CbfWorkspace(d, Bn, dev)
        for _ in range(3):
            torch_api.cbf_solve_dev(d, *a, ws=ws)
        tm = torch_api.Timer(); tm.begin()
        for _ in range(10):
            torch_api.cbf_solve_dev(d, *a, ws=ws)
        tm.end(); ms = tm.ms() / 10
        st, it = ws.status.cpu().numpy(), ws.iters.cpu().numpy()
        print("%s N=%2d: %.3f ms per %d NLPs = %.3g /s   converged %.1f %%   iterations mean %.1f max %d   (%.2f us per problem-iteration)" % (
            kind, N, ms, Bn, Bn / ms * 1e3, 100 * (st == 0).mean(), it.mean(), it.max(), ms * 1e3 / it.sum() * 1.0))
