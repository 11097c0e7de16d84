"""Per-iteration scalars (e_d, e_p, e_c, mu, alpha, alpha_d, dw, accepted) of one problem of the descriptor fuzz (tests/test_gpu_parity.py
test_fuzz_descriptors) -- for A/B-ing two builds (CRX_LIB=...) when they disagree.  python tools/trace_fuzz.py TRIAL INDEX"""
import os, sys, ctypes as C
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd"]
import numpy as np
import crx
from crx import abi, synth
gpu = crx.init(); L = crx.lib(); A, B = synth.load_AB()
T, I = int(sys.argv[1]), int(sys.argv[2])
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
one = [a[I:I + 1] for a in args]
if os.environ.get("CRX_MAXIT"): d.opts.max_iter = int(os.environ["CRX_MAXIT"])
if os.environ.get("CRX_POISON"): L.crx_debug_poison_lds(1)
L.crx_trace_enable(0, 64)
r = gpu.cbf_solve(d, *one)
buf = np.zeros((64, 16)); L.crx_trace_read(buf.ctypes.data_as(C.c_void_p), 64)
L.crx_trace_enable(0, 0)
if os.environ.get("CRX_TRACE_DUMP"):   # rows 32..63: whatever an instrumented build left there (tools/ab/bugsrc/dump)
    np.save(os.environ["CRX_TRACE_DUMP"], buf)
print("trial %d problem %d N %d V %d n_obs %d: iters %d status %d kkt %.3e" % (T, I, N, V, int(one[5][0]), int(r["iters"][0]), int(r["status"][0]), float(r["kkt"][0])))
print("  X", np.array2string(np.asarray(r["X"])[0].ravel()[:24], precision=17), "\n  U", np.array2string(np.asarray(r["U"])[0].ravel()[:8], precision=17), "\n  sumX %.17e sumU %.17e" % (np.asarray(r["X"]).sum(), np.asarray(r["U"]).sum()))
for i in range(min(int(r["iters"][0]) + 1, 64)):
    print("  %2d e_d %.17e e_p %.17e e_c %.6e mu %.3e al %.17e a_d %.6e dw %.1e acc %d" % ((i,) + tuple(buf[i, :7]) + (int(buf[i, 7]),)))
    if os.environ.get("CRX_TRACE_COLS"):   # columns 8..15: phase clocks of a -DCRX_PHASE_CLOCKS build, or whatever an instrumented build put there
        print("       " + " ".join("%.17e" % v for v in buf[i, 8:16]))
