R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg3; do
 for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc/${wl}_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc/${wl}_$ctr -o p -- python $R/bench.py --steps 10 --warmup 2 --workload $wl --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc/err_${wl}_$ctr.log
 done
done
find $R/gpurun_out/pmc -type f | head -30
