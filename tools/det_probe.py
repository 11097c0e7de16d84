import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/car-racing_amd"); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import crx, helpers
from crx import montecarlo, synth, abi
from utils import racing_env
A, B = synth.load_AB()
track = racing_env.ClosedTrack(np.genfromtxt(ROOT + "/data/track_layout/l_shape.csv", delimiter=","), track_width=1.0)
tab, L = track.point_and_tangent, track.lap_length
gpu = crx.init()
def races():
    n = 64; z = np.zeros((n, 6))
    return montecarlo.mpccbf_races(tab, L, track.width, A, B, z, z, np.tile([4.0, 10.0], (n, 1)), np.tile([0.2, 0.2], (n, 1)), np.tile([0.1, -0.1], (n, 1)), 400)
def report(tag, r):
    x = r["xcurv"]; same = (x == x[:, :1]).all(axis=(0, 2))
    first = [int(np.argmax((x[:, b] != x[:, 0]).any(axis=1))) for b in range(x.shape[1]) if not same[b]]
    print(tag, "identical races:", int(same.sum()), "/ 64; first differing step:", sorted(first)[:5], "nan:", np.isnan(x).any())
r0 = races(); report("fresh      ", r0)
which = sys.argv[1] if len(sys.argv) > 1 else "lmpc"
if which in ("lmpc", "all"):
    g = np.load(ROOT + "/tests/golden/racing_game.npz"); d, args = helpers.lmpc_inputs(g)
    big = [np.concatenate([a] * 20) for a in args]; gpu.lmpc_solve(d, *big)
    r1 = races(); report("after lmpc ", r1)
if which in ("cfg4", "all"):
    p = synth.cfg4_tracking_cbf(2048, N=20)
    dd = abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True)
    gpu.cbf_solve(dd, p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], p["n_obs"])
    r2 = races(); report("after cfg4 ", r2)
if which in ("prep", "all"):
    pp = synth.cfg3_planner(1024, N=12); w = pp["raw"]
    dp = abi.prep_desc(12, 3, len(w["opt_s"]), w["track_width"], w["lap_length"])
    gpu.planner_prep(dp, w["x"], w["x"], w["n_veh"], w["veh_info"], w["max_dv"], w["obs_s"], w["obs_ey"], w["opt_s"], w["opt_ey"])
    r3 = races(); report("after prep ", r3)
r4 = races(); report("again      ", r4)
print("run-to-run identical:", np.array_equal(r0["xcurv"], r4["xcurv"]))
