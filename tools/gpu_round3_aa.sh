# round 3, GPU call AA: probe -- horizon as a compile-time constant (N = NMAX) in the solver kernels
R=$GRAFT_REPO_ROOT
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged_frac']))"; }
for rep in 1 2; do
for v in intree nfix; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "$v cfg2"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "$v cfg3"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg3 --batch 16384 --no-cpu-baseline --steps 20 --warmup 3 2> /dev/null | line "$v cfg3x16384"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | line "$v cfg4"
done
done
