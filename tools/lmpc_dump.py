"""Diagnostics: solve the recorded learning-MPC QPs one by one (batch 1, as the closed loop does) and dump status /
iterations / u0 to gpurun_out/lmpc_dump_<tag>.npz (A/B a kernel change: run with two builds, compare)."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/car-racing_amd"); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import crx, helpers
g = np.load(ROOT + "/tests/golden/racing_game.npz")
gpu = crx.init()
d, args = helpers.lmpc_inputs(g)
n = args[0].shape[0]
st, it, u0, kkt = [], [], [], []
for i in range(n):
    r = gpu.lmpc_solve(d, *[a[i:i + 1] for a in args])
    st.append(int(r["status"][0])); it.append(int(r["iters"][0])); u0.append(r["U"][0, :2].copy()); kkt.append(float(r["kkt"][0]))
rb = gpu.lmpc_solve(d, *args)
os.makedirs(ROOT + "/gpurun_out", exist_ok=True)
np.savez(ROOT + "/gpurun_out/lmpc_dump_%s.npz" % sys.argv[1], st=st, it=it, u0=np.array(u0), kkt=kkt, stb=rb["status"], itb=rb["iters"], ub=rb["U"])
print("status", st); print("iters ", it); print("batch status", list(rb["status"]))
