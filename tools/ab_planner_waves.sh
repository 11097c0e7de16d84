# A/B: third resident wave per SIMD for the planner instantiation (waves_per_eu 3, 21 spilled dwords) vs the default
R=$GRAFT_REPO_ROOT
cd $R
run() { python bench.py --no-cpu-baseline "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' '.join(sys.argv[1:]), '| value %.4g ms/step %.4g kernel_ms %.4g resident %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['resident_problems_per_cu']))" "$@"; }
for v in 1 3 2; do
  make -C car-racing_amd/csrc -s clean; make -C car-racing_amd/csrc -s EXTRA=-DCRX_PLANNER_WAVES=$v 2>&1 | grep -E "error" 
  echo "== CRX_PLANNER_WAVES=$v"
  run --workload cfg3 --steps 100 --warmup 10
  run --workload cfg3 --batch 16384 --steps 20 --warmup 3
  run --workload cfg5 --steps 20 --warmup 3
done
make -C car-racing_amd/csrc -s clean; make -C car-racing_amd/csrc -s
