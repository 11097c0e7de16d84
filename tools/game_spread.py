"""Natural spread of the batched racing-game lap with the reference's traffic: 32 copies of the scenario whose start
state differs by 1e-9 (rounding-sized) noise.  The loop is chaotic; the spread over the copies is what bounds a test that
compares two runs of "the same" scenario.  Usage (GPU box): python tools/game_spread.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "car-racing_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import helpers
    import scenarios
    from crx import montecarlo, synth
    from utils import racing_env
    A, B = synth.load_AB()
    g = np.load(os.path.join(ROOT, "tests", "golden", "racing_game.npz"))
    track = racing_env.ClosedTrack(np.genfromtxt(os.path.join(ROOT, "data/track_layout/l_shape.csv"), delimiter=","), track_width=1.0)
    opt = scenarios.table("optimal_traj", "xcurv_l_shape")
    d, ss, us, qf, time_ss, lin_points, lin_input = helpers.lmpc_lap_setup(g, track)
    cars = scenarios.RACING_GAME["cars"]
    Bn, steps = 32, 200
    rng = np.random.default_rng(3)
    s0 = np.tile([c[1] for c in cars], (Bn, 1)); v = np.tile([c[2] for c in cars], (Bn, 1)); ey = np.tile([c[3] for c in cars], (Bn, 1))
    x0 = np.tile(g["lmpc/x"][0], (Bn, 1)); xg0 = np.tile(g["lap1/xglob"][-1], (Bn, 1))
    x0[1:, [0, 5]] += rng.normal(0, 1e-9, (Bn - 1, 2))
    tile = lambda a: np.tile(a[None], (Bn,) + (1,) * a.ndim)   # noqa: E731
    r = montecarlo.game_laps(track.point_and_tangent, track.lap_length, track.width, A, B, opt, tile(ss), tile(us), tile(qf), tile(time_ss),
                             np.full(Bn, 2, dtype=np.int32), x0, xg0, tile(lin_points), tile(lin_input), s0, v, ey, steps)
    x = r["xcurv"]
    L = track.lap_length
    done = np.array([int(np.nonzero(np.diff(x[:, b, 4]) < -5.0)[0][0]) + 1 if (np.diff(x[:, b, 4]) < -5.0).any() else -1 for b in range(Bn)])
    print("lap time (steps): min %d max %d" % (done[done > 0].min(), done.max()), "unfinished", int((done < 0).sum()))
    n = done[done > 0].min()
    ds = np.abs(x[:n, :, 4] - x[:n, :1, 4]); ds = np.minimum(ds, np.abs(ds - L))
    de = np.abs(x[:n, :, 5] - x[:n, :1, 5])
    for t in (15, 30, 60, 90, n - 1):
        print("step %3d: s spread max %.3e  ey spread max %.3e" % (t, ds[t].max(), de[t].max()))
    print("whole lap: s spread max %.3f  ey spread max %.3f" % (ds.max(), de.max()))
    print("overtake steps per race:", r["overtake"][:n].sum(axis=0).min(), "..", r["overtake"][:n].sum(axis=0).max())


if __name__ == "__main__":
    main()
