# round 3, GPU call W: terminal-set reachability screen of the learning-MPC QP's first attempt: suite, then game with / without
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/suite_w.log 2>&1; grep -E "passed|failed|rror" gpurun_out/suite_w.log | tail -5
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f  iters p50 %s p90 %s max %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], c['converged_frac'], c['iters_p50'], c['iters_p90'], c['iters_max']))"; }
for rep in 1 2; do
for f in "" "--no-reach-screen"; do
  timeout 300 python bench.py --workload game $f --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "game $f"
  timeout 300 python bench.py --workload lmpc $f --no-cpu-baseline --steps 50 --warmup 5 2> /dev/null | line "lmpc $f"
done
done
