"""Bit-for-bit A/B of the planner instantiation alone (a cut of tools/cbf_ab.py): [CRX_LIB=...] python tools/plan_ab.py TAG; --compare A B"""
import os, sys
import numpy as np
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd", ROOT + "/tests"]
OUT = ROOT + "/gpurun_out/plan_ab_%s.npz"
if sys.argv[1] == "--compare":
    a, b = np.load(OUT % sys.argv[2]), np.load(OUT % sys.argv[3])
    for k in a.files:
        if not np.array_equal(a[k], b[k], equal_nan=True):
            d = np.abs(a[k].astype(float) - b[k].astype(float))
            print("  %s differs: max %.3g, %d of %d entries" % (k, np.nanmax(d), int((d > 0).sum()), d.size))
    print("compared", sys.argv[2], sys.argv[3])
    sys.exit(0)
import crx
from crx import abi, synth
gpu = crx.init()
A, B = synth.load_AB()
out = {}
for nm, n, N, seed in (("cfg3", 1024, 12, 3), ("plan10", 256, 10, 5), ("plan20", 256, 20, 9)):
    p = synth.cfg3_planner(n, N=N, seed=seed)
    for qm in (0, 1):
        d = abi.planner_desc(N, A, B)
        d.opts.qp_method = qm
        r = gpu.planner_solve(d, *[p[k] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")])
        for k, v in r.items():
            out["%s/q%d/%s" % (nm, qm, k)] = np.asarray(v)
np.savez(OUT % sys.argv[1], **out)
print(sys.argv[1], "done")
