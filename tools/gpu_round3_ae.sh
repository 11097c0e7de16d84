# round 3, GPU call AE: machine-scheduler strategies again, on the fixed-horizon / fixed-exponent instantiations
R=$GRAFT_REPO_ROOT
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for v in intree pl_itilp pl_maxilp pl_maxmem; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "$v cfg3"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --steps 20 --warmup 3 2> /dev/null | line "$v cfg5"
done
for v in intree ob_def ob_maxilp; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "$v cfg2"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | line "$v cfg4"
  CRX_LIB=$lib timeout 300 python bench.py --workload races --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "$v races"
done
done
