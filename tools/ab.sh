cd $GRAFT_REPO_ROOT
for wl in ${WLS:-cfg2 cfg3 cfg4 lmpc races}; do
  python bench.py --workload $wl --no-cpu-baseline --steps ${STEPS:-50} --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['roofline']['resident_problems_per_cu'])"
done
