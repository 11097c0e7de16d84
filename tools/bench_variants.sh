# the bench.py variants DESIGN.md section 6 quotes besides the five workloads of tools/gpu_prof.sh
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print(' '.join(sys.argv[1:]), '| value %.4g ms/step %.4g conv %.4f iters p50/max %s/%s host_call_ms %s resident %s' % (d['value'], d['ms_per_step'], c['converged_frac'], c['iters_p50'], c['iters_max'], c['p50_host_call_one_control_step_ms'], d['roofline']['resident_problems_per_cu']))" "$@"; }
run --workload cfg3 --device-prep --steps 50 --warmup 5
run --workload cfg2 --no-scenario-filter --steps 50 --warmup 5
run --workload cfg3 --batch 16384 --steps 10 --warmup 3
run --workload cfg2 --batch 16384 --steps 10 --warmup 3
run --workload races --steps 100 --warmup 5
