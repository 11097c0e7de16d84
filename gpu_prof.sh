set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg3; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/$wl -o $wl -- python $R/bench.py --steps 50 --warmup 5 --workload $wl --no-cpu-baseline > $R/gpurun_out/prof/bench_$wl.json 2> $R/gpurun_out/prof/err_$wl.log
done
ls -R $R/gpurun_out/prof | head -40
