# round-2 profiles: rocprofv3 --kernel-trace --stats of bench.py for every workload -> profiles/r02_*_kernel_stats.txt
# (summaries are written on the GPU box into gpurun_out/prof2/ and copied to profiles/ by hand after review)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg2_filtered cfg3 cfg4 lmpc cfg5 races game overtake; do
  st=50; [ $wl = cfg4 ] && st=5; [ $wl = lmpc ] && st=10; [ $wl = races ] && st=30; [ $wl = cfg5 ] && st=8; [ $wl = game ] && st=40; [ $wl = overtake ] && st=40
  ex=""
  rm -rf $O/$wl
  rocprofv3 --kernel-trace --stats -d $O/$wl -o $wl -- python $R/bench.py --steps $st --warmup 3 --workload $wl $ex --no-cpu-baseline > $O/bench_$wl.json 2> $O/err_$wl.log
  db=$(find $O/$wl -name "*.db" | head -1)
  python3 $R/profiles/summarize.py $db $O/bench_$wl.json > $O/r02_${wl}_kernel_stats.txt 2>> $O/err_$wl.log
  find $O/$wl -type f ! -name "*.txt" -delete
done
ls -la $O | head -30
