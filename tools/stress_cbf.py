"""Kernel vs oracle on LARGE draws of the MPC-CBF NLPs (the parity tests hold 256 / 192 problems): statuses, iteration counts and costs of
16 384 cfg2 and 8 192 cfg4 problems, several seeds.  python tools/stress_cbf.py [n_seeds]"""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd", ROOT + "/tests"]
import numpy as np
import crx, oracle, kkt_check
from crx import abi, synth
gpu = crx.init(); orc = oracle.load(); A, B = synth.load_AB()
KEYS = ("x0", "xt", "obs_s", "obs_ey", "lap_off", "n_obs")
for seed in range(1, 1 + (int(sys.argv[1]) if len(sys.argv) > 1 else 2)):
    for name, p, d in (("cfg2", synth.cfg2_mpccbf(16384, seed=seed, safe_start=False), abi.cbf_desc(12, 1, A, B, alpha=0.8, margin=0.2)),
                       ("cfg4", synth.cfg4_tracking_cbf(8192, seed=seed + 10, safe_start=False), abi.cbf_desc(20, 3, A, B, alpha=0.6, margin=0.15, Q=(10.0, 0, 0, 5.0, 0, 50.0), per_stage_target=True))):
        args = [p[k] for k in KEYS]
        t0 = time.time(); rg = gpu.cbf_solve(d, *args); tg = time.time() - t0
        t0 = time.time(); ro = orc.cbf_solve(d, *args); to = time.time() - t0
        sg, so, ig, io = rg["status"], ro["status"], rg["iters"], ro["iters"]
        both = (sg == 0) & (so == 0)
        rel = np.abs(rg["cost"][both] - ro["cost"][both]) / np.maximum(1.0, np.abs(ro["cost"][both]))
        dx = np.abs(rg["X"][both] - ro["X"][both]).reshape(both.sum(), -1).max(axis=1)
        print("seed %d %s: n %d | converged gpu %d oracle %d, one side only %d (gpu only %d, oracle only %d) | same status %.4f same iters %.4f |iters diff| max %d | both converged: cost rel max %.1e (> 1e-6: %d), |dX| max %.1e (> 1e-5: %d) | kkt max gpu %.1e | %.1f s gpu call, %.1f s oracle" % (
            seed, name, len(sg), (sg == 0).sum(), (so == 0).sum(), ((sg == 0) != (so == 0)).sum(), ((sg == 0) & (so != 0)).sum(), ((sg != 0) & (so == 0)).sum(), (sg == so).mean(), (ig == io).mean(), np.abs(ig - io).max(),
            rel.max(), (rel > 1e-6).sum(), dx.max(), (dx > 1e-5).sum(), rg["kkt"][sg == 0].max(), tg, to), flush=True)
        # every both-converged pair that differs: the oracle-free certificate on both points (tests/kkt_check.py)
        lines, table = kkt_check.classify_pairs(d, p, rg, ro)
        print("   pairs with |dX| > 1e-5: %s" % table)
        for ln in lines:
            if "two KKT" in ln or "NOT" in ln:
                print("     " + ln)
        for b in np.nonzero(sg != so)[0]:
            print("     status differs #%d: gpu %d after %d, oracle %d after %d" % (b, sg[b], ig[b], so[b], io[b]))
