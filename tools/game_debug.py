"""Debug aid: the batched racing-game lap with traffic (crx.montecarlo.game_laps) against the class-surface run, step by step."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (ROOT, ROOT + "/car-racing_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np
import conftest, helpers, scenarios
from control import lmpc_helper
from crx import montecarlo, synth
A, B = synth.load_AB()
g = np.load(ROOT + "/tests/golden/racing_game.npz")
track = scenarios.make_track("l_shape", 1.0)
opt = scenarios.table("optimal_traj", "xcurv_l_shape")
d, ss, us, qf, time_ss, lin_points, lin_input = helpers.lmpc_lap_setup(g, track)
cars = scenarios.RACING_GAME["cars"]
Bn, steps = 4, 160
s0 = np.tile([c[1] for c in cars], (Bn, 1)); v = np.tile([c[2] for c in cars], (Bn, 1)); ey = np.tile([c[3] for c in cars], (Bn, 1))
x0 = np.tile(g["lmpc/x"][0], (Bn, 1)); xg0 = np.tile(g["lap1/xglob"][-1], (Bn, 1))
tile = lambda a: np.tile(a[None], (Bn,) + (1,) * a.ndim)
r = montecarlo.game_laps(track.point_and_tangent, track.lap_length, track.width, A, B, opt, tile(ss), tile(us), tile(qf), tile(time_ss),
                         np.full(Bn, 2, dtype=np.int32), x0, xg0, tile(lin_points), tile(lin_input), s0, v, ey, steps)
lmpc_helper.ON_SINGULAR = "keep"
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    race, ctrl = scenarios.racing_game(dict(scenarios.RACING_GAME, lap_plan=("pid", "mpc-lti", "lmpc+traffic")))
one = np.array(race.ego.xcurvs[2]); uone = np.array(race.ego.inputs[2])
ot_one = [p is None for p in race.ego.lmpc_prediction]     # True where the overtake branch ran
x = r["xcurv"][:, 0]
for k in range(min(len(one) - 1, steps)):
    print("k %3d batched ot %d flag %2d | surface ot %d | dx %.2e du %.2e | s %.3f %.3f vx %.3f %.3f ey %.3f %.3f" % (
        k, r["overtake"][k, 0], r["flag"][k, 0], ot_one[k] if k < len(ot_one) else -1, np.abs(x[k] - one[k]).max(),
        np.abs(r["u"][k, 0] - uone[k]).max() if k < len(uone) else -1, x[k, 4], one[k, 4], x[k, 0], one[k, 0], x[k, 5], one[k, 5]))
