# round-2 HBM traffic of the solver kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg3 cfg4 cfg5; do
 st=10; [ $wl = cfg4 ] && st=3; [ $wl = cfg5 ] && st=3
 for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${wl}_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${wl}_$ctr -o p -- python $R/bench.py --steps $st --warmup 2 --workload $wl --no-cpu-baseline > /dev/null 2> $O/err_${wl}_$ctr.log
 done
done
python3 $R/profiles/summarize_pmc.py r02
