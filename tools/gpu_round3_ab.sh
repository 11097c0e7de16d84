# round 3, GPU call AB: fixed-horizon instantiations (crx_solve_kernel<NOBS, NMAX, DEG, NFIX>) against -DCRX_NFIX=0: bit-identity, suite, speed
R=$GRAFT_REPO_ROOT
cd $R
CRX_LIB=$R/tools/ab/libcrx_nfix0.so python tools/cbf_ab.py nfix0 2>&1 | tail -1; python tools/cbf_ab.py nfix 2>&1 | tail -1; python tools/cbf_ab.py --compare nfix0 nfix
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/suite_ab.log 2>&1; grep -E "passed|failed|rror" gpurun_out/suite_ab.log | tail -5
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged_frac']))"; }
for v in intree nfix0; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  for wl in cfg2 cfg3 cfg4 cfg5 races overtake; do
    st=30; [ $wl = cfg2 -o $wl = cfg3 ] && st=200; [ $wl = cfg4 ] && st=10; [ $wl = overtake ] && st=60
    CRX_LIB=$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --steps $st --warmup 5 2> /dev/null | line "$v $wl"
  done
done
