# round 3, GPU call Q: closed-loop workloads on libcrx's measured streams (crx_streams_create): sub-batches 1 / 2 / 4, hardware queues
# 4 / 8, same steps as the default bench; then the closed-loop GPU tests and the default bench line (all workloads in one process)
R=$GRAFT_REPO_ROOT
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g steps/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"; }
for q in 4 8; do
  for wl in races game overtake; do
    st=30; [ $wl != races ] && st=60
    for k in 1 2 4; do
      GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --workload $wl --race-streams $k --no-cpu-baseline --steps $st --warmup 5 2> /dev/null | line "queues $q $wl sub-batches $k"
    done
  done
done
timeout 900 python -m pytest tests/test_gpu_closed_loop.py -q -x -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
python bench.py > gpurun_out/bench_default_q.json 2> gpurun_out/bench_default_q.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_default_q.json").read().strip().splitlines()[-1])
for k, v in d["summary"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "kernel_ms", "converged_frac", "iters_max", "dispatch")})
PY
