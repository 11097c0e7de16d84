cd $GRAFT_REPO_ROOT && make -C oracle -s 2>&1 | grep -i error
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k lmpc -x 2>&1 | tail -15
timeout 120 python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "car-racing_amd"); sys.path.insert(0, "tests")
import crx, oracle, helpers
g = np.load("tests/golden/racing_game.npz")
gpu = crx.init(); orc = oracle.load()
d, args = helpers.lmpc_inputs(g)
rg = gpu.lmpc_solve(d, *args); ro = orc.lmpc_solve(d, *args)
print("gpu status", rg["status"]); print("orc status", ro["status"])
print("gpu iters", rg["iters"]); print("orc iters", ro["iters"])
ok = g["lmpc_success"]
print("dX gpu-golden", np.abs(rg["X"][ok] - g["lmpc/X"][ok]).max(), "gpu-orc", np.abs(rg["X"][ok] - ro["X"][ok]).max())
big = [np.concatenate([a] * 64) for a in args]
gpu.lmpc_solve(d, *big)
t0 = time.time(); r = gpu.lmpc_solve(d, *big); t1 = time.time()
print("batch", len(big[0]), "time %.3f ms -> %.0f solves/s" % ((t1 - t0) * 1e3, len(big[0]) / (t1 - t0)))
one = [a[:1] for a in args]
gpu.lmpc_solve(d, *one)
t0 = time.time()
for _ in range(20): gpu.lmpc_solve(d, *one)
print("single-problem blocking call %.3f ms" % ((time.time() - t0) / 20 * 1e3))
PY
