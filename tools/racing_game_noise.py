"""Diagnostics: how robust is the racing-game scenario (tests/test_gpu_closed_loop.py::test_racing_game) to
perturbations at solver-tolerance level?  Every learning-MPC input gets noise of the given size; prints the outcome per seed."""
import os, sys, io, contextlib
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/car-racing_amd"); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import crx
import control.control as cc
import test_gpu_closed_loop as t
orig = crx.lmpc_solve
from control import lmpc_helper
if len(sys.argv) > 3: lmpc_helper.ON_SINGULAR = sys.argv[3]
size = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-10
class Cap:
    def __init__(self): self.buf = io.StringIO()
    def readouterr(self):
        class O: pass
        o = O(); o.out = self.buf.getvalue(); return o
for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    rng = np.random.default_rng(seed)
    calls = [0, 0]
    last = []
    def hooked(d, *args, **kw):
        r = orig(d, *args, **kw)
        r["U"] = r["U"] + size * rng.standard_normal(r["U"].shape)
        calls[0] += 1; calls[1] += int(r["status"][0] != 0)
        last.append("call %d st %d it %d kkt %.1e max|X| %s max|U| %s" % (calls[0], r["status"][0], r["iters"][0], r["kkt"][0],
                    np.array2string(np.abs(r["X"][0]).max(axis=0), precision=2), np.array2string(np.abs(r["U"][0]).max(axis=0), precision=2)))
        del last[:-6]
        return r
    crx.lmpc_solve = hooked
    if hasattr(cc, "crx"): cc.crx.lmpc_solve = hooked
    cap = Cap()
    try:
        with contextlib.redirect_stdout(cap.buf):
            t.test_racing_game(cap)
        res = "passed"
    except BaseException as e:
        res = "%s %s" % (type(e).__name__, str(e)[:80].replace("\n", " "))
    print("seed", seed, "noise", size, "lmpc calls", calls[0], "not converged", calls[1], "->", res, flush=True)
    if res != "passed":
        for l in last: print("seed", seed, "   ", l, flush=True)
