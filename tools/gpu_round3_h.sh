# round 3, GPU call H: why are side streams slower?  (a) races: direct on the current stream vs one side stream; (b) GPU_MAX_HW_QUEUES
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3h
mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "car-racing_amd")]
import numpy as np, torch
import crx
from crx import montecarlo, synth
from utils import racing_env
crx.init(0)
A, B = synth.load_AB()
track = racing_env.ClosedTrack(np.genfromtxt("data/track_layout/l_shape.csv", delimiter=","), track_width=1.0)
rng = np.random.default_rng(50)
n = 4096
s0 = np.sort(rng.uniform(3.0, 17.0, (n, 2)), axis=1); s0[:, 1] = np.maximum(s0[:, 1], s0[:, 0] + 2.0)
cv, ce = rng.uniform(0.1, 0.4, (n, 2)), rng.choice([-0.5, -0.3, -0.1, 0.1, 0.3, 0.5], (n, 2))
def mk(sl):
    m = sl.stop - sl.start
    return montecarlo.MpccbfRaces(track.point_and_tangent, track.lap_length, track.width, A, B, np.zeros((m, 6)), np.zeros((m, 6)), s0[sl], cv[sl], ce[sl], vt=0.8, N=10)
def timeit(stepf, tag, steps=60):
    for _ in range(5): stepf()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): stepf()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    print("%-50s %.4f ms/step" % (tag, dt)); sys.stdout.flush()
r = mk(slice(0, n)); timeit(r.step, "one batch, current (default) stream")
st = torch.cuda.Stream()
r = mk(slice(0, n))
def side():
    with torch.cuda.stream(st): r.step()
timeit(side, "one batch, one side stream")
def side_join():
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st): r.step()
    torch.cuda.current_stream().wait_stream(st)
timeit(side_join, "one batch, side stream, fork/join per step")
for k in (2, 4):
    cuts = [slice(i * n // k, (i + 1) * n // k) for i in range(k)]
    c = montecarlo.Concurrent([mk(sl) for sl in cuts])
    timeit(c.step, "%d sub-batches on %d side streams (free-running)" % (k, k))
    parts = [mk(sl) for sl in cuts]
    def seq():
        for p in parts: p.step()
    timeit(seq, "%d sub-batches, all on the current stream" % k)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
PY
for q in 4 8; do
  for wl in races overtake; do
    GPU_MAX_HW_QUEUES=$q python bench.py --workload $wl --race-streams 2 --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$q $wl streams 2: %.4g steps/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"
  done
done
