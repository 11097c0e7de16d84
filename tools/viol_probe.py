"""GPU diagnostics: the CBF rows of every converged cfg4 trajectory recomputed from the outputs; the worst one next to the oracle's
solve of the same problem."""
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd"]
import crx   # noqa: E402
import oracle   # noqa: E402
from crx import abi, synth   # noqa: E402

gpu = crx.init(); orc = oracle.load()
A, B = synth.load_AB()
Bn = 16384
p = synth.cfg4_tracking_cbf(Bn, N=20, seed=4, safe_start=False)
d = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
keys = ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")
r = gpu.cbf_solve(d, *[p[k] for k in keys])
ok = r["status"] == 0
X, sg = r["X"][ok], r["sigma"][ok]
al, cm = 0.6, 1.15
de = (X[:, None, :, 5] - p["obs_ey"][ok]) / 0.2
dn = (X[:, None, :, 4] - p["obs_s"][ok]) / 0.4
dc = (X[:, None, :, 4] - p["obs_s"][ok] - p["lap_off"][ok][:, :, None]) / 0.4
hn = dn ** 6 + de ** 6 - cm - sg; hc = dc ** 6 + de ** 6 - cm - sg
row = hn[:, :, 1:] - (1 - al) * hc[:, :, :-1]
present = np.arange(3)[None, :] < p["n_obs"][ok][:, None]
viol = np.where(present[:, :, None], row, np.inf)
bad = np.nonzero(viol.min(axis=(1, 2)) < -1e-6)[0]
print("converged", int(ok.sum()), "with a violated row:", len(bad), "min", viol.min())
for i in bad[:3]:
    gi = int(np.nonzero(ok)[0][i])
    o, k = np.unravel_index(np.argmin(viol[i]), viol[i].shape)
    print("problem", gi, "obstacle", o, "stage", k, "row", viol[i, o, k], "iters", r["iters"][gi], "kkt", r["kkt"][gi], "cost", r["cost"][gi],
          "n_obs", p["n_obs"][gi], "lap_off", p["lap_off"][gi])
    print("  sigma gpu", sg[i, o, :8])
    ro = orc.cbf_solve(d, *[p[kk][gi:gi + 1] for kk in keys])
    print("  oracle status", ro["status"][0], "iters", ro["iters"][0], "cost", ro["cost"][0], "sigma", ro["sigma"][0][o][:8])
    print("  rows gpu", row[i, o, :8])
    print("  x0", p["x0"][gi], "obs_s", p["obs_s"][gi][o][:3], "obs_ey", p["obs_ey"][gi][o][:3])
