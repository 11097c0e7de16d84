# round 3, GPU call Z: the exponent of the super-ellipse as a compile-time constant (crx_solve_kernel<NOBS, NMAX, 6>) against the
# general instantiation (-DCRX_DEG6=0): bit-identity on the parity suites, then speed
R=$GRAFT_REPO_ROOT
cd $R
CRX_LIB=$R/tools/ab/libcrx_deg0.so python tools/cbf_ab.py deg0 2>&1 | tail -1; python tools/cbf_ab.py deg6 2>&1 | tail -1; python tools/cbf_ab.py --compare deg0 deg6
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/suite_z.log 2>&1; grep -E "passed|failed|rror" gpurun_out/suite_z.log | tail -5
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged_frac']))"; }
for rep in 1 2; do
for v in intree deg0; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "$v cfg2"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg2 --batch 4096 --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "$v cfg2x4096"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | line "$v cfg4"
  CRX_LIB=$lib timeout 300 python bench.py --workload races --race-streams 1 --no-cpu-baseline --steps 50 --warmup 5 2> /dev/null | line "$v races K=1"
done
done
