cd $GRAFT_REPO_ROOT
run() { python bench.py --workload $1 --batch $2 --no-cpu-baseline --steps $3 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 batch $2', d['value'], d['ms_per_step'], d['roofline']['resident_problems_per_cu'])"; }
run cfg3 16384 10
run cfg2 16384 10
run cfg2 4096 20
run races 16384 10
