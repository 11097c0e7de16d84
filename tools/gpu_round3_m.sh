# round 3, GPU call M: scheduler-strategy builds of crx_lmpc.hip + crx_lmpcprep.hip
R=$GRAFT_REPO_ROOT
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2; do
for v in intree lm_itilp lm_maxilp lm_maxmem; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload lmpc --no-cpu-baseline --steps 100 --warmup 10 2> /dev/null | line "$v lmpc"
  CRX_LIB=$lib timeout 300 python bench.py --workload game --race-streams 1 --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "$v game K=1"
done
done
