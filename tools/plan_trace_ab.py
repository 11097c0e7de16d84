"""Per-iteration trace scalars of ONE planner QP of the cfg3 draw, saved bit-exact for A/B-ing two builds: [CRX_LIB=...] python tools/plan_trace_ab.py TAG IDX [QM]; --compare A B"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd"]
OUT = ROOT + "/gpurun_out/plan_trace_%s.npy"
if sys.argv[1] == "--compare":
    a, b = np.load(OUT % sys.argv[2]), np.load(OUT % sys.argv[3])
    names = "ed ep ec mu al a_d dw acc s8 s9 s10 s11 s12 s13 s14 s15".split()
    for it in range(len(a)):
        bad = [(names[j], a[it, j], b[it, j]) for j in range(16) if a[it, j].tobytes() != b[it, j].tobytes()]
        if bad:
            print("first difference at iteration", it, ["%s %.17g %.17g" % x for x in bad]); break
    else:
        print("traces identical")
    sys.exit(0)
import crx
from crx import abi, synth
gpu, L = crx.init(), crx.lib()
A, B = synth.load_AB()
idx = int(sys.argv[2]); qm = int(sys.argv[3]) if len(sys.argv) > 3 else 0
p = synth.cfg3_planner(1024, N=12, seed=3)
d = abi.planner_desc(12, A, B); d.opts.qp_method = qm
a = tuple(p[k][idx:idx + 1] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub"))
L.crx_trace_enable(0, 64)
r = gpu.planner_solve(d, *a)
buf = np.zeros((64, 16))
L.crx_trace_read(buf.ctypes.data_as(C.c_void_p), 64)
L.crx_trace_enable(0, 0)
np.save(OUT % sys.argv[1], buf)
print(sys.argv[1], "status", r["status"][0], "iters", r["iters"][0])
for it in range(int(r["iters"][0]) + 1):
    t = buf[it]; print("it %2d ed %.17g ep %.17g mu %.3g al %.17g a_d %.17g" % (it, t[0], t[1], t[3], t[4], t[5]))
