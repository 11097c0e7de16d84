# round 3, GPU call S: reachability screen of the planner QPs -- GPU suite, then cfg3 / cfg5 / overtake with and without it
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/suite_s.log 2>&1; grep -E "passed|failed|rror" gpurun_out/suite_s.log | tail -5
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f  iters p50 %s max %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], c['converged_frac'], c['iters_p50'], c['iters_max']))"; }
for f in "" "--no-reach-screen"; do
  timeout 300 python bench.py --workload cfg3 $f --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "cfg3 $f"
  timeout 300 python bench.py --workload cfg3 --batch 16384 $f --no-cpu-baseline --steps 20 --warmup 3 2> /dev/null | line "cfg3x16384 $f"
  timeout 300 python bench.py --workload cfg5 $f --no-cpu-baseline --steps 20 --warmup 3 2> /dev/null | line "cfg5 weak $f"
  timeout 300 python bench.py --workload cfg5 --scaling strong $f --no-cpu-baseline --steps 5 --warmup 2 2> /dev/null | line "cfg5 strong $f"
  timeout 300 python bench.py --workload overtake $f --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "overtake $f"
done
