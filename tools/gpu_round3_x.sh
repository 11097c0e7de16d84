# round 3, GPU call X: first attempt also skipped when the fixed x_0 violates its own rows: suite, overtake / game with and without the screens
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/suite_x.log 2>&1; grep -E "passed|failed|rror" gpurun_out/suite_x.log | tail -5
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], c['converged_frac']))"; }
for rep in 1 2; do
for f in "" "--no-reach-screen"; do
  timeout 300 python bench.py --workload overtake $f --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "overtake $f"
  timeout 300 python bench.py --workload game $f --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "game $f"
done
done
python tools/multi_tests.py 4096 400 3 1 1 2>&1 | grep "race-steps"
