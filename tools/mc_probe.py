import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/car-racing_amd")
import numpy as np
from crx import montecarlo, synth
from utils import racing_env
A, B = synth.load_AB()
track = racing_env.ClosedTrack(np.genfromtxt(ROOT + "/data/track_layout/l_shape.csv", delimiter=","), track_width=1.0)
tab, L = track.point_and_tangent, track.lap_length
rng = np.random.default_rng(5)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
s0 = np.sort(rng.uniform(3.0, 17.0, (m, 2)), axis=1); s0[:, 1] = np.maximum(s0[:, 1], s0[:, 0] + 2.0)
v = rng.uniform(0.1, 0.4, (m, 2)); ey = rng.choice([-0.5, -0.3, -0.1, 0.1, 0.3, 0.5], (m, 2))
montecarlo.mpccbf_races(tab, L, track.width, A, B, np.zeros((8, 6)), np.zeros((8, 6)), s0[:8], v[:8], ey[:8], 5)
t0 = time.time()
rr = montecarlo.mpccbf_races(tab, L, track.width, A, B, np.zeros((m, 6)), np.zeros((m, 6)), s0, v, ey, steps, vt=0.8)
el = time.time() - t0
x = rr["xcurv"]; t = np.arange(1, x.shape[0]) * 0.1
print("races %d steps %d wall %.2f s -> %.0f race-steps/s; converged %.4f" % (m, steps, el, m * steps / el, (rr["status"] == 0).mean()))
hmin = np.full(m, np.inf)
for c in range(2):
    so = v[None, :, c] * t[:, None] + s0[None, :, c]
    ds = (x[1:, :, 4] - so + 0.5 * L) % L - 0.5 * L
    h = (ds / 0.4) ** 6 + ((x[1:, :, 5] - ey[None, :, c]) / 0.2) ** 6
    hmin = np.minimum(hmin, h.min(axis=0))
clean = (rr["status"] == 0).all(axis=0)
ego_total = x[1:, :, 4] + L * np.cumsum(np.diff(x[:, :, 4], axis=0) < -0.5 * L, axis=0)
print("ey max", np.abs(x[:, :, 5]).max(), "width", track.width)
worst = np.inf
for c in range(2):
    so = v[None, :, c] * t[:, None] + s0[None, :, c]
    away = (ego_total < L - 3.0) & (so < L - 3.0)
    h = ((ego_total - so) / 0.4) ** 6 + ((x[1:, :, 5] - ey[None, :, c]) / 0.2) ** 6
    hm = np.where(away & clean[None, :], h, np.inf)
    k, b = np.unravel_index(np.argmin(hm), hm.shape)
    print("car", c, "worst away-from-line h %.4f at step %d race %d: ego_total %.3f ey %.3f car s %.3f ey %.3f; races with h<0.9: %d" % (hm[k, b], k, b, ego_total[k, b], x[k + 1, b, 5], so[k, b], ey[b, c], (hm.min(axis=0) < 0.9).sum()))
print("races with contact (h<1): %.3f ; among all-converged races (%d): %.3f" % ((hmin < 1).mean(), clean.sum(), (hmin[clean] < 1).mean()))
print("h min quantiles", np.quantile(hmin, [0, 0.01, 0.05, 0.25, 0.5]))
print("progress s+laps*L: mean %.2f" % (x[-1, :, 4] + rr["laps"] * L).mean(), "laps", np.bincount(rr["laps"]))
bad = np.argsort(hmin)[:3]
for b in bad: print("worst", b, "s0", s0[b], "v", v[b], "ey", ey[b], "hmin", hmin[b])
