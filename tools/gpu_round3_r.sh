# round 3, GPU call R: rocprofv3 kernel stats of the closed-loop workloads on the final defaults (4 sub-batches, libcrx streams)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof3
mkdir -p $P
for wl in races game overtake; do
  st=30; [ $wl != races ] && st=40
  rm -rf $P/$wl
  rocprofv3 --kernel-trace --stats -d $P/$wl -o $wl -- python $R/bench.py --steps $st --warmup 3 --workload $wl --no-cpu-baseline > $P/bench_$wl.json 2> $P/err_$wl.log
  db=$(find $P/$wl -name "*.db" | head -1)
  python3 $R/profiles/summarize.py $db $P/bench_$wl.json > $P/r03_${wl}_kernel_stats.txt 2>> $P/err_$wl.log
  find $P/$wl -type f ! -name "*.txt" -delete
  head -16 $P/r03_${wl}_kernel_stats.txt | cut -c1-110
done
