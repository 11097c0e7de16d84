# round 3, GPU call T: learning-MPC kernel at six per CU (LDS 31 664 -> 27 248 B): bit-identity against the previous build, then speed
R=$GRAFT_REPO_ROOT
cd $R
CRX_LIB=$R/tools/ab/libcrx_base.so python tools/lmpc_ab.py base 2>&1 | grep -v amdgpu | tail -1
python tools/lmpc_ab.py new 2>&1 | grep -v amdgpu | tail -1
python tools/lmpc_ab.py --compare base new
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  resident %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['resident_problems_per_cu']))"; }
for rep in 1 2; do
for v in base intree; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib timeout 300 python bench.py --workload lmpc --no-cpu-baseline --steps 100 --warmup 10 2> /dev/null | line "$v lmpc"
  CRX_LIB=$lib timeout 300 python bench.py --workload game --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "$v game"
  CRX_LIB=$lib timeout 300 python bench.py --workload overtake --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "$v overtake"
done
done
