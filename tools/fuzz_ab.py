"""Outputs of one trial of the descriptor fuzz (tests/test_gpu_parity.py::test_fuzz_descriptors) on the loaded library, dumped for a bit-for-bit
A/B of two builds: [CRX_LIB=...] python tools/fuzz_ab.py TRIAL TAG;  python tools/fuzz_ab.py --compare TRIAL TAG1 TAG2"""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd"]
import numpy as np
OUT = ROOT + "/gpurun_out/fuzz_ab_%s_%s.npz"
if sys.argv[1] == "--compare":
    a, b = np.load(OUT % (sys.argv[2], sys.argv[3])), np.load(OUT % (sys.argv[2], sys.argv[4]))
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        msg = "identical" if same else "DIFFERENT in problems %s" % np.unique(np.nonzero(~np.isclose(a[k], b[k], rtol=0, atol=0, equal_nan=True))[0]).tolist()
        print("  %-8s %s" % (k, msg))
    print("  iters", a["iters"].tolist()); print("  iters", b["iters"].tolist()); print("  oracle", a["o_iters"].tolist())
    sys.exit(0)
import crx, oracle
from crx import abi, synth
gpu = crx.init(); orc = oracle.load(); A, B = synth.load_AB()
T = int(sys.argv[1])
rng = np.random.default_rng(2024)
for trial in range(14):
    N = int(rng.integers(3, 25)); V = int(rng.integers(0, 4)); nb = 48
    if V == 0:
        p = synth.cfg2_mpccbf(nb, N=N, seed=100 + trial, n_obs=1)
        d = abi.cbf_desc(N, 0, A, B, Q=tuple(rng.uniform(0, 30, 6)), R=tuple(rng.uniform(0.05, 1.0, 2)), ey_max=float(rng.uniform(0.7, 1.2)))
        args = (p["x0"], p["xt"], np.zeros((nb, 0, N + 1)), np.zeros((nb, 0, N + 1)), np.zeros((nb, 0)), np.zeros(nb, np.int32))
    else:
        per_stage = bool(rng.integers(0, 2))
        p = synth.cfg4_tracking_cbf(nb, N=N, seed=100 + trial, n_obs=V) if per_stage else synth.cfg2_mpccbf(nb, N=N, seed=100 + trial, n_obs=V)
        d = abi.cbf_desc(N, V, A, B, alpha=float(rng.uniform(0.3, 1.0)), margin=float(rng.uniform(0.05, 0.3)), degree=int(rng.choice([2, 4, 6])), per_stage_target=per_stage,
                         Q=(10.0, 0, 0, float(rng.uniform(1, 8)), 0, float(rng.uniform(10, 60))))
        n = rng.integers(0, V + 1, nb).astype(np.int32)
        args = (p["x0"], p["xt"], p["obs_s"], p["obs_ey"], p["lap_off"], n)
    d.opts.restore_iters = -1
    if trial == T:
        break
r, ro = gpu.cbf_solve(d, *args), orc.cbf_solve(d, *args)
np.savez(OUT % (T, sys.argv[2]), X=r["X"], U=r["U"], status=r["status"], iters=r["iters"], kkt=r["kkt"], cost=r["cost"], o_iters=ro["iters"], o_status=ro["status"], o_cost=ro["cost"])
print("trial %d N %d V %d: dumped" % (T, N, V))
