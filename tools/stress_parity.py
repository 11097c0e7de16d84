"""One-off parity stress on the GPU box: a full cfg5 shard (16384 scenarios = 65536 region QPs, seed of choice) through kernel
AND oracle, every status / iteration count compared (the pytest suite compares subsamples), plus HiGHS on a sample of the
verdicts.  usage: python tools/stress_parity.py [seed=101] [n_scen=16384]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "car-racing_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import crx
    import oracle
    from crx import abi, synth
    from scipy.optimize import linprog
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 101
    n_scen = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    A, B = synth.load_AB()
    N = 12
    p = synth.cfg3_planner(n_scen, N=N, seed=seed)
    d = abi.planner_desc(N, A, B)
    args = [p[k] for k in ("x0", "bez_s", "bez_ey", "ey_lb", "ey_ub")]
    gpu = crx.init(); orc = oracle.load()
    t0 = time.time(); rg = gpu.planner_solve(d, *args); t1 = time.time(); ro = orc.planner_solve(d, *args); t2 = time.time()
    sg, so, ig, io = rg["status"], np.asarray(ro["status"]), rg["iters"], np.asarray(ro["iters"])
    print("%d QPs: GPU %.2f s (host call), oracle %.2f s (%d threads)" % (len(sg), t1 - t0, t2 - t1, oracle.threads()))
    print("status GPU %s oracle %s" % (np.bincount(sg, minlength=3), np.bincount(so, minlength=3)))
    ds = np.nonzero((sg != 0) != (so != 0))[0]
    di = np.nonzero((sg == so) & (ig != io))[0]
    print("verdict mismatches: %d ; same verdict, different iteration count: %d (max |diff| %d)" % (len(ds), len(di), np.abs(ig[di] - io[di]).max() if len(di) else 0))
    for b in ds[:10]:
        print("   verdict", b, "gpu", sg[b], ig[b], "oracle", so[b], io[b])
    ok = (sg == 0) & (so == 0)
    print("converged on both: max |X diff| %.2e  max |U diff| %.2e" % (np.abs(rg["X"][ok] - np.asarray(ro["X"])[ok]).max(), np.abs(rg["U"][ok] - np.asarray(ro["U"])[ok]).max()))
    # LP check of a sample of GPU verdicts
    Ap = [np.eye(6)]
    for _ in range(N):
        Ap.append(A @ Ap[-1])
    G = np.zeros((N + 1, 6, 2 * N))
    for k in range(1, N + 1):
        for j in range(k):
            G[k][:, 2 * j:2 * j + 2] = Ap[k - 1 - j] @ B
    bounds = [(-d.delta_max, d.delta_max), (-d.a_max, d.a_max)] * N
    rng = np.random.default_rng(seed)
    wrong = 0
    sample = rng.choice(len(sg), 2000, replace=False)
    for b in sample:
        x0, lb, ub = p["x0"][b], p["ey_lb"][b], p["ey_ub"][b]
        rows, rhs = [], []
        for k in range(1, N + 1):
            f = Ap[k] @ x0
            rows.append(G[k][0]); rhs.append(d.vx_max - f[0])
            if k < N:
                if np.isfinite(ub):
                    rows.append(G[k][5]); rhs.append(ub - f[5])
                if np.isfinite(lb[k]):
                    rows.append(-G[k][5]); rhs.append(f[5] - lb[k])
        infeas0 = x0[5] < lb[0] - 1e-8 or x0[5] > ub + 1e-8
        st = [linprog(np.zeros(2 * N), A_ub=np.array(rows), b_ub=np.array(rhs) + m, bounds=bounds, method="highs").status == 2 for m in (0.0, -1e-7, 1e-7)]
        if (sg[b] != 0) != (st[0] or infeas0) and st[1] == st[2]:
            wrong += 1
    print("HiGHS on %d sampled GPU verdicts: %d wrong" % (len(sample), wrong))


if __name__ == "__main__":
    main()
