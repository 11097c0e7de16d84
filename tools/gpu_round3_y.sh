# round 3, GPU call Y: straight products for the integer powers 6 / 5 / 4 of the super-ellipse (CRX_IPOW_FAST) against the select chain
R=$GRAFT_REPO_ROOT
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged_frac']))"; }
for rep in 1 2; do
for v in intree ipow0; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "$v cfg2"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg2 --batch 4096 --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "$v cfg2x4096"
  CRX_LIB=$lib timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | line "$v cfg4"
  CRX_LIB=$lib timeout 300 python bench.py --workload races --race-streams 1 --no-cpu-baseline --steps 50 --warmup 5 2> /dev/null | line "$v races K=1"
done
done
CRX_LIB=$R/tools/ab/libcrx_ipow0.so python tools/lmpc_ab.py base > /dev/null 2>&1
python - <<PY
import sys, numpy as np
sys.path[:0] = ["$R", "$R/car-racing_amd"]
import os
PY
