# round 3, GPU call AF: slack start of the CBF NLPs (provable lower bounds): GPU suite, then cfg2 / cfg4 / races / overtake with and without
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/suite_af.log 2>&1; grep -E "passed|failed|rror" gpurun_out/suite_af.log | tail -5
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: %.4g /s (converged %.4g)  %.4f ms/step  status %s  iters p50 %s p90 %s max %s' % (d['value'], d['value_converged'], d['ms_per_step'], {k: round(v, 4) for k, v in c['status_frac'].items() if v}, c['iters_p50'], c['iters_p90'], c['iters_max']))"; }
for f in "" "--slack-start"; do
  timeout 300 python bench.py --workload cfg2 $f --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "cfg2 $f"
  timeout 300 python bench.py --workload cfg2 --batch 4096 $f --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "cfg2x4096 $f"
  timeout 300 python bench.py --workload cfg4 $f --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | line "cfg4 $f"
  timeout 300 python bench.py --workload races $f --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "races $f"
  timeout 300 python bench.py --workload overtake $f --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "overtake $f"
done
