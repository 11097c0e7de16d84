"""The reference's Monte-Carlo experiment (car_racing/tests/overtake_planner_test.py --multi-tests: 100 racing games against
random scripted cars, one after the other) as ONE batched, device-resident run: B races x `steps` control steps of
crx.montecarlo.GameLaps against crx.synth.multi_tests_traffic.  Prints what such a sweep is run for: lap times of the
learning-MPC laps, overtakes, contacts, races that left the track.
The plant runs WITH the reference's bounded process noise (utils/base.py:929-939), as the reference's script does unless
--zero-noise is given (overtake_planner_test.py:41-42): crx_plant_step_noise_dev, draws from a seeded torch generator.
usage (GPU box): python tools/multi_tests.py [B=4096] [steps=400] [num_veh=3] [noise_seed=1 | none] [sub-batches=1]
CRX_OPTS="max_iter=3000,restore_iters=3000,reach_screen=0" in the environment overrides crx_ipm_opts fields of EVERY descriptor the run
builds (crx.abi.OPTS_OVERRIDE) -- tools/budget_experiment.sh uses it to ask whose collisions these are: the controller's or the
solver budgets' (VERDICT r3 item 5)."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime initialises: GameLaps' two branch streams need their own hardware queues

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "car-racing_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    from crx import abi, montecarlo, synth
    from utils import racing_env
    for kv in filter(None, os.environ.get("CRX_OPTS", "").split(",")):
        k, v = kv.split("=")
        abi.OPTS_OVERRIDE[k] = float(v) if k in ("tol", "mu_init", "kappa_eps", "kappa_mu", "theta_mu", "tau_min", "slack_push", "grad_scale_max") else int(v)
    if abi.OPTS_OVERRIDE:
        print("crx_ipm_opts overrides:", abi.OPTS_OVERRIDE)
    Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    V = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    seed = None if (len(sys.argv) > 4 and sys.argv[4].lower() in ("none", "zero")) else (int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    A, B = synth.load_AB()
    g = np.load(os.path.join(ROOT, "tests", "golden", "racing_game.npz"))
    track = racing_env.ClosedTrack(np.genfromtxt(os.path.join(ROOT, "data/track_layout/l_shape.csv"), delimiter=","), track_width=1.0)
    opt = np.genfromtxt(os.path.join(ROOT, "data/optimal_traj/xcurv_l_shape.csv"), delimiter=",")
    N, L = 12, track.lap_length
    ss = np.ascontiguousarray(g["ss/ss0"].transpose(2, 0, 1)); us = np.ascontiguousarray(g["ss/u0"].transpose(2, 0, 1))
    qf = np.ascontiguousarray(g["ss/Qfun0"].T); time_ss = g["ss/time_ss"].astype(np.int32)
    x0 = np.tile(g["lmpc/x"][0], (Bn, 1)); xg0 = np.tile(g["lap1/xglob"][-1], (Bn, 1))
    s0, v, ey = synth.multi_tests_traffic(Bn, V, seed=1)
    K = int(sys.argv[5]) if len(sys.argv) > 5 else 1      # concurrent sub-batches (crx.montecarlo.Concurrent); the per-step statistics below are ~15 small torch launches per sub-batch, which makes this loop launch-bound beyond K = 1..2 (bench.py --workload overtake, without them, gains from 4)
    cuts = [(Bn * i) // K for i in range(K + 1)]
    parts = []
    for i in range(K):
        sl = slice(cuts[i], cuts[i + 1]); n = sl.stop - sl.start
        tl = lambda a: np.tile(a[None], (n,) + (1,) * a.ndim)   # noqa: E731
        parts.append(montecarlo.GameLaps(track.point_and_tangent, L, track.width, A, B, opt, tl(ss), tl(us), tl(qf), tl(time_ss),
                                         np.full(n, 2, dtype=np.int32), x0[sl], xg0[sl], tl(ss[0, 1:N + 2]), tl(us[0, 1:N + 1]), s0[sl], v[sl], ey[sl],
                                         noise_seed=None if seed is None else seed + i))
    conc = montecarlo.Concurrent(parts)
    dev = parts[0].lm.xc.device
    # per-race statistics stay on the device and are updated on the sub-batch's own stream: no host synchronisation inside the loop
    st = []
    for i, r in enumerate(parts):
        sl = slice(cuts[i], cuts[i + 1]); n = sl.stop - sl.start
        st.append(dict(cs0=torch.as_tensor(s0[sl], device=dev), cv=torch.as_tensor(v[sl], device=dev), cey=torch.as_tensor(ey[sl], device=dev),
                       prev=r.lm.laps.clone(), ncross=torch.zeros(n, dtype=torch.int32, device=dev),
                       t1=torch.zeros(n, dtype=torch.int32, device=dev), t2=torch.zeros(n, dtype=torch.int32, device=dev),
                       gap=torch.full((n,), 1e9, dtype=torch.float64, device=dev), off=torch.zeros(n, dtype=torch.bool, device=dev),
                       eymax=torch.zeros(n, dtype=torch.float64, device=dev), ot=torch.zeros(n, dtype=torch.int64, device=dev)))
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(steps):
        for r, q, stream in zip(parts, st, conc.streams):
            with torch.cuda.stream(stream):
                r.step()
                cars = q["cv"] * r.t + q["cs0"]
                ds = torch.remainder(r.lm.xc[:, 4:5] - cars + 0.5 * L, L) - 0.5 * L
                de = r.lm.xc[:, 5:6] - q["cey"]
                q["gap"] = torch.minimum(q["gap"], ((ds / 0.4) ** 6 + (de / 0.2) ** 6).min(dim=1).values)   # >= 1: no contact
                q["off"] |= r.lm.xc[:, 5].abs() > track.width
                q["eymax"] = torch.maximum(q["eymax"], r.lm.xc[:, 5].abs())
                q["ot"] += r.overtake.long()
                crossed = r.lm.laps > q["prev"]
                q["t1"] = torch.where(crossed & (q["ncross"] == 0), k + 1, q["t1"])
                q["t2"] = torch.where(crossed & (q["ncross"] == 1), k + 1, q["t2"])
                q["ncross"] += crossed.int()
                q["prev"] = r.lm.laps.clone()
    torch.cuda.synchronize(); wall = time.time() - t0
    cat = lambda key: torch.cat([q[key] for q in st]).cpu().numpy()   # noqa: E731
    laps, t1, t2 = cat("ncross"), cat("t1"), cat("t2")
    first, second = t1[laps >= 1], (t2 - t1)[laps >= 2]
    mg, off, ey_max, ot_steps = cat("gap"), torch.cat([q["off"] for q in st]), torch.cat([q["eymax"] for q in st]), torch.cat([q["ot"] for q in st])
    xc_all = torch.cat([r.lm.xc for r in parts])
    print("%d races x %d steps against %d random cars each, process noise %s, %d concurrent sub-batches: %.2f s wall = %.3g race-steps/s" % (
        Bn, steps, V, "ON (seed %d)" % seed if seed is not None else "off", K, wall, Bn * steps / wall))
    print("laps completed per race: %s" % dict(zip(*np.unique(laps, return_counts=True))))
    if len(first):
        print("first learning-MPC lap : steps p5 %d p50 %d p95 %d" % tuple(np.percentile(first, [5, 50, 95])))
    if len(second):
        print("second learning-MPC lap: steps p5 %d p50 %d p95 %d" % tuple(np.percentile(second, [5, 50, 95])))
    print("steps in the overtake branch per race: p5 %d p50 %d p95 %d" % tuple(np.percentile(ot_steps.cpu().numpy(), [5, 50, 95])))
    print("contact with a car (super-ellipse (ds/0.4)^6 + (dey/0.2)^6 < 1 at some step): %d races (%.1f %%)" % ((mg < 1.0).sum(), 100 * (mg < 1.0).mean()))
    print("left the track (|ey| > %.1f at some step): %d races (%.1f %%)" % (track.width, off.sum().item(), 100 * off.float().mean().item()))
    print("max |ey| over the run: p50 %.3f p90 %.3f p99 %.3f max %.3f" % tuple(np.percentile(ey_max.cpu().numpy(), [50, 90, 99, 100])))
    print("non-finite states: %d" % int((~torch.isfinite(xc_all)).any(dim=1).sum().item()))


if __name__ == "__main__":
    main()
