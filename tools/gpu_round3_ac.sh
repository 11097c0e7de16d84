# round 3, GPU call AC: probe -- horizon (and safe-set size) as compile-time constants in crx_lmpc_kernel
R=$GRAFT_REPO_ROOT
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged_frac']))"; }
for rep in 1 2; do
for v in intree lmN lmNM; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload lmpc --no-cpu-baseline --steps 50 --warmup 5 2> /dev/null | line "$v lmpc"
  CRX_LIB=$lib timeout 300 python bench.py --workload game --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "$v game"
done
done
