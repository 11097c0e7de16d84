"""Diagnostics: run the racing-game scenario of tests/test_gpu_closed_loop.py with every crx.lmpc_solve call logged
(inputs, status, iterations, u0) to gpurun_out/racing_game_log_<tag>.npz."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/car-racing_amd"); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import crx
log = []
orig = crx.lmpc_solve
def hooked(d, *args, **kw):
    r = orig(d, *args, **kw)
    log.append(dict(x=np.array(args[0]).copy(), st=int(r["status"][0]), it=int(r["iters"][0]), u0=r["U"][0].ravel()[:2].copy(), kkt=float(r["kkt"][0])))
    return r
crx.lmpc_solve = hooked
import control.control as cc
if hasattr(cc, "crx"): cc.crx.lmpc_solve = hooked
import test_gpu_closed_loop as t
class Cap:
    def readouterr(self):
        class O: out = ""
        return O()
try:
    t.test_racing_game(Cap())
    print("test passed")
except BaseException as e:
    print("test raised", type(e).__name__, str(e)[:200])
os.makedirs(ROOT + "/gpurun_out", exist_ok=True)
np.savez(ROOT + "/gpurun_out/racing_game_log_%s.npz" % sys.argv[1], x=np.array([l["x"].ravel() for l in log]), st=[l["st"] for l in log],
         it=[l["it"] for l in log], u0=np.array([l["u0"] for l in log]), kkt=[l["kkt"] for l in log])
print(len(log), "lmpc calls; status counts", np.bincount([l["st"] for l in log]))
