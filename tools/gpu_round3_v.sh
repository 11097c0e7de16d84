# round 3, GPU call V: patience of the learning-MPC stagnation rule (iterations with mu < 1e-6 before a QP is left on its noise floor)
R=$GRAFT_REPO_ROOT
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  status %s  iters p50 %s p90 %s max %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], {k: round(v, 4) for k, v in c['status_frac'].items()}, c['iters_p50'], c['iters_p90'], c['iters_max']))"; }
for v in intree late16 late12 late8; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload game --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "$v game"
  CRX_LIB=$lib timeout 300 python bench.py --workload game --no-cpu-baseline --steps 200 --warmup 5 2> /dev/null | line "$v game 200 steps"
  CRX_LIB=$lib timeout 300 python bench.py --workload overtake --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "$v overtake"
  CRX_LIB=$lib timeout 300 python bench.py --workload lmpc --no-cpu-baseline --steps 50 --warmup 5 2> /dev/null | line "$v lmpc"
done
