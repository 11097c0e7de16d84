"""Kernel vs oracle on the general instantiations, compactly: for N in argv (default 24) and V = 0, 1, 2, 3, 5 a 256-problem draw with restore_iters = -1;
prints the problems whose status / iteration count leaves the oracle's.  [CRX_LIB=...] python tools/general_check.py [N ...]"""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd"]
import numpy as np
import crx, oracle
from crx import abi, synth
gpu = crx.init(); orc = oracle.load(); A, B = synth.load_AB()
for N in [int(a) for a in sys.argv[1:]] or [24]:
    for V in (0, 1, 2, 3, 5):
        Bn = 256
        if V == 0:
            p = synth.cfg2_mpccbf(Bn, N=N, seed=500 + N, n_obs=1); d = abi.cbf_desc(N, 0, A, B)
            args = (p["x0"], p["xt"], np.zeros((Bn, 0, N + 1)), np.zeros((Bn, 0, N + 1)), np.zeros((Bn, 0)), np.zeros(Bn, np.int32))
        else:
            ps = V >= 2
            p = synth.cfg4_tracking_cbf(Bn, N=N, seed=500 + N + V, n_obs=V) if ps else synth.cfg2_mpccbf(Bn, N=N, seed=500 + N + V, n_obs=V)
            d = abi.cbf_desc(N, V, A, B, alpha=0.6 if ps else p.get("alpha", 0.6), margin=0.15 if ps else p.get("margin", 0.15), per_stage_target=ps, **({"Q": (10.0, 0, 0, 5.0, 0, 50.0)} if ps else {}))
            n = np.random.default_rng(N * 10 + V).integers(0, V + 1, Bn).astype(np.int32); n[: Bn // 2] = V
            args = (p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], n)
        d.opts.restore_iters = -1
        rg, ro = gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)
        bad = np.nonzero((rg["status"] != ro["status"]) | (rg["iters"] != ro["iters"]))[0]
        both = (rg["status"] == 0) & (ro["status"] == 0) & (rg["iters"] == ro["iters"])
        dX = np.abs(rg["X"][both] - ro["X"][both]).max() if both.any() else float("nan")
        print("N=%d V=%d: %3d of %d off the oracle %s  n_obs of those %s  max|dX| on the agreeing ones %.1e" % (
            N, V, len(bad), Bn, bad[:12].tolist(), args[5][bad[:12]].tolist(), dX))
