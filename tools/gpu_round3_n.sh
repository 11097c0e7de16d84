# round 3, GPU call N: the two-translation-unit build (obstacle instantiations with iterative-ilp scheduling): full GPU suite, then
# A/B of obstacle-unit variants (no 2-wave floor for <1,12>; MachineLICM on)
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged_frac']))"; }
for rep in 1 2; do
for v in intree o1w1 olicm; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "$v cfg2"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg2 --batch 4096 --dispatch index --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "$v cfg2x4096 index"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | line "$v cfg4 auto"
  CRX_LIB=$lib timeout 300 python bench.py --workload races --no-cpu-baseline --steps 50 --warmup 5 2> /dev/null | line "$v races"
done
done
