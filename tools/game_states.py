"""States of the benched learning-MPC loop (bench.py `game`: LmpcLaps from the reference's recorded safe set, perturbed starts) at several
lap phases, dumped for the reference-side fixture generator (tests/golden/tools/make_draws.py game -> tests/golden/game_draw.npz):
    python tools/game_states.py [n_races] [phase ...]       (GPU box; writes gpurun_out/game_states.npz)
Per phase p and race: the state BEFORE control step p (x, u_old, linearisation points / inputs, safe set, cost-to-go, lap bookkeeping) and what
the device made of it in step p (stage models, selected safe-set points, QP status / iterations, the applied input)."""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, ROOT + "/car-racing_amd"]
import numpy as np
import torch
import crx
from crx import montecarlo
from utils import racing_env

Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 32
phases = [int(a) for a in sys.argv[2:]] or [0, 20, 40, 60]
crx.init(0); dev = torch.device("cuda", 0)
g = np.load(os.path.join(ROOT, "tests", "golden", "racing_game.npz"))
track = racing_env.ClosedTrack(np.genfromtxt(os.path.join(ROOT, "data/track_layout/l_shape.csv"), delimiter=","), track_width=1.0)
N = 12
ss = np.ascontiguousarray(g["ss/ss0"].transpose(2, 0, 1)); us = np.ascontiguousarray(g["ss/u0"].transpose(2, 0, 1))
qf = np.ascontiguousarray(g["ss/Qfun0"].T); time_ss = g["ss/time_ss"].astype(np.int32)
rng = np.random.default_rng(60)                     # bench.py make_game, rank 0: the same perturbed starts (its first Bn races)
x0 = np.tile(g["lmpc/x"][0], (4096, 1)); xg0 = np.tile(g["lap1/xglob"][-1], (4096, 1))
x0[:, 0] += rng.uniform(-0.03, 0.03, 4096); x0[:, 5] += rng.uniform(-0.05, 0.05, 4096); xg0[:, 0] = x0[:, 0]
x0, xg0 = x0[:Bn], xg0[:Bn]
t = lambda a: np.tile(a[None], (Bn,) + (1,) * a.ndim)   # noqa: E731
r = montecarlo.LmpcLaps(track.point_and_tangent, track.lap_length, track.width, t(ss), t(us), t(qf), t(time_ss), np.full(Bn, 2, dtype=np.int32),
                        x0, xg0, t(ss[0, 1:N + 2]), t(us[0, 1:N + 1]), N=N, device=dev)
out = {"phases": np.array(phases), "lap_length": track.lap_length, "track_width": track.width}
c = lambda a: a.detach().cpu().numpy().copy()   # noqa: E731
for k in range(max(phases) + 1):
    if k in phases:
        torch.cuda.synchronize()
        pre = dict(x=c(r.xc), u_old=c(r.u_old), lin_points=c(r.lin_points), lin_input=c(r.lin_input), ss=c(r.ss), us=c(r.us), qf=c(r.qf),
                   time_ss=c(r.time_ss), it=c(r.it), step_no=c(r.step_no))
    r.step()
    if k in phases:
        torch.cuda.synchronize()
        post = dict(A=c(r.pws.A), B=c(r.pws.B), C=c(r.pws.C), ss_sel=c(r.pws.ss), q_sel=c(r.pws.qfun), status=c(r.ws.status), iters=c(r.ws.iters),
                    U=c(r.ws.U), X=c(r.ws.X))
        for n, v in {**pre, **post}.items():
            out["p%d/%s" % (k, n)] = v
        st = post["status"]
        print("phase %3d: QP status counts %s   iterations mean %.1f" % (k, np.bincount(st, minlength=6).tolist(), post["iters"].mean()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "game_states.npz"), **out)
print("wrote gpurun_out/game_states.npz")
