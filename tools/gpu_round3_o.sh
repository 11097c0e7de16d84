# round 3, GPU call O: wavefronts per race in crx_lmpc_prep_kernel (LP_WAVES = 4 in-tree; 6, 8, 12, 16)
R=$GRAFT_REPO_ROOT
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for v in intree lp6 lp8 lp12 lp16; do
  lib=$R/tools/ab/libcrx_$v.so; [ $v = intree ] && lib=$R/car-racing_amd/crx/libcrx.so
  CRX_LIB=$lib python tools/prep_probe.py 30 2>&1 | grep crx_lmpc_prep
  CRX_LIB=$lib timeout 300 python bench.py --workload game --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "$v game K=2"
  CRX_LIB=$lib timeout 300 python bench.py --workload overtake --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "$v overtake K=2"
done
CRX_LIB=$R/tools/ab/libcrx_lp12.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_closed_loop.py -q -x -m gpu -k "lmpc or game or closed or laps" 2>&1 | grep -E "passed|failed"
