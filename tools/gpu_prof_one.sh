# usage: gpu_prof_one.sh <workload> <steps>: rocprofv3 kernel stats of one bench workload -> gpurun_out/prof2/r02_<wl>_kernel_stats.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
wl=$1; st=$2
mkdir -p $O
ex=""
cd /tmp && export TMPDIR=/tmp
rm -rf $O/$wl
rocprofv3 --kernel-trace --stats -d $O/$wl -o $wl -- python $R/bench.py --steps $st --warmup 3 --workload $wl $ex --no-cpu-baseline > $O/bench_$wl.json 2> $O/err_$wl.log
db=$(find $O/$wl -name "*.db" | head -1)
python3 $R/profiles/summarize.py $db $O/bench_$wl.json > $O/r02_${wl}_kernel_stats.txt 2>> $O/err_$wl.log
find $O/$wl -type f ! -name "*.txt" -delete
grep "crx_" $O/r02_${wl}_kernel_stats.txt | cut -c1-170
