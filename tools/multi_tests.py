"""The reference's Monte-Carlo experiment (car_racing/tests/overtake_planner_test.py --multi-tests: 100 racing games against
random scripted cars, one after the other) as ONE batched, device-resident run: B races x `steps` control steps of
crx.montecarlo.GameLaps against crx.synth.multi_tests_traffic.  Prints what such a sweep is run for: lap times of the
learning-MPC laps, overtakes, contacts, races that left the track.
The plant runs WITH the reference's bounded process noise (utils/base.py:929-939), as the reference's script does unless
--zero-noise is given (overtake_planner_test.py:41-42): crx_plant_step_noise_dev, draws from a seeded torch generator.
usage (GPU box): python tools/multi_tests.py [B=4096] [steps=400] [num_veh=3] [noise_seed=1 | none]"""
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
    from crx import montecarlo, synth
    from utils import racing_env
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
    tile = lambda a: np.tile(a[None], (Bn,) + (1,) * a.ndim)   # noqa: E731
    r = montecarlo.GameLaps(track.point_and_tangent, L, track.width, A, B, opt, tile(ss), tile(us), tile(qf), tile(time_ss),
                            np.full(Bn, 2, dtype=np.int32), x0, xg0, tile(ss[0, 1:N + 2]), tile(us[0, 1:N + 1]), s0, v, ey, noise_seed=seed)
    dev = r.lm.xc.device
    cs0, cv, cey = (torch.as_tensor(a, device=dev) for a in (s0, v, ey))
    lap_steps = [[] for _ in range(Bn)]
    prev_laps = r.lm.laps.clone()
    min_gap = torch.full((Bn,), 1e9, dtype=torch.float64, device=dev)       # min over time and cars of (ds/l)^6 + (dey/w)^6 (>= 1: no contact)
    off = torch.zeros((Bn,), dtype=torch.bool, device=dev)
    ey_max = torch.zeros((Bn,), dtype=torch.float64, device=dev)
    ot_steps = torch.zeros((Bn,), dtype=torch.int64, device=dev)
    ahead0 = None
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(steps):
        r.step()
        cars = cv * r.t + cs0
        ds = torch.remainder(r.lm.xc[:, 4:5] - cars + 0.5 * L, L) - 0.5 * L
        de = r.lm.xc[:, 5:6] - cey
        min_gap = torch.minimum(min_gap, ((ds / 0.4) ** 6 + (de / 0.2) ** 6).min(dim=1).values)
        off |= r.lm.xc[:, 5].abs() > track.width
        ey_max = torch.maximum(ey_max, r.lm.xc[:, 5].abs())
        ot_steps += r.overtake.long()
        crossed = (r.lm.laps > prev_laps).cpu().numpy(); prev_laps = r.lm.laps.clone()
        for b in np.nonzero(crossed)[0]:
            lap_steps[b].append(k + 1)
    torch.cuda.synchronize(); wall = time.time() - t0
    laps = np.array([len(x) for x in lap_steps])
    first = np.array([x[0] for x in lap_steps if len(x) >= 1]); second = np.array([x[1] - x[0] for x in lap_steps if len(x) >= 2])
    mg = min_gap.cpu().numpy()
    print("%d races x %d steps against %d random cars each, process noise %s: %.2f s wall = %.3g race-steps/s" % (
        Bn, steps, V, "ON (seed %d)" % seed if seed is not None else "off", wall, Bn * steps / wall))
    print("laps completed per race: %s" % dict(zip(*np.unique(laps, return_counts=True))))
    if len(first):
        print("first learning-MPC lap : steps p5 %d p50 %d p95 %d" % tuple(np.percentile(first, [5, 50, 95])))
    if len(second):
        print("second learning-MPC lap: steps p5 %d p50 %d p95 %d" % tuple(np.percentile(second, [5, 50, 95])))
    print("steps in the overtake branch per race: p5 %d p50 %d p95 %d" % tuple(np.percentile(ot_steps.cpu().numpy(), [5, 50, 95])))
    print("contact with a car (super-ellipse (ds/0.4)^6 + (dey/0.2)^6 < 1 at some step): %d races (%.1f %%)" % ((mg < 1.0).sum(), 100 * (mg < 1.0).mean()))
    print("left the track (|ey| > %.1f at some step): %d races (%.1f %%)" % (track.width, off.sum().item(), 100 * off.float().mean().item()))
    print("max |ey| over the run: p50 %.3f p90 %.3f p99 %.3f max %.3f" % tuple(np.percentile(ey_max.cpu().numpy(), [50, 90, 99, 100])))
    print("non-finite states: %d" % int((~torch.isfinite(r.lm.xc)).any(dim=1).sum().item()))


if __name__ == "__main__":
    main()
