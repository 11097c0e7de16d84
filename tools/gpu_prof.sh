set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
make -C $R/oracle -s
cd $R && python -m pytest tests -m gpu -q 2>&1 | grep -v "solver time\|^overtaking\|local planner" | tail -3
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg3 cfg4 lmpc races; do
  st=50; [ $wl = cfg4 ] && st=5; [ $wl = lmpc ] && st=10; [ $wl = races ] && st=30
  rm -rf $R/gpurun_out/prof/$wl
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/$wl -o $wl -- python $R/bench.py --steps $st --warmup 5 --workload $wl --no-cpu-baseline > $R/gpurun_out/prof/bench_$wl.json 2> $R/gpurun_out/prof/err_$wl.log
done
cd $R && python bench.py > gpurun_out/prof/bench_default.json; cat gpurun_out/prof/bench_default.json | cut -c1-200
