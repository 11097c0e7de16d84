# round 3, GPU call P (final build of the round: two translation units, dispatch order): the suite, the default bench line, rocprofv3 kernel stats of every workload, HBM-traffic PMC passes
# (FETCH_SIZE / WRITE_SIZE) with the kernel-source hash, the issue / wait counters, slim vs full <3,20> layout.
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3p
mkdir -p $O $R/gpurun_out/pmc3 $R/gpurun_out/prof3
make -C oracle -s
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -25 ) > $O/pytest_gpu.log
grep -h "passed\|failed" $O/pytest_gpu.log | tail -2
python -c "import bench; print(bench.kernel_source_hash())" > $R/gpurun_out/pmc3/source_hash.txt   # sha over crx_kernels.hip, crx_wave.h, Makefile
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["value_converged"], d["ms_per_step"])
for k, v in d["summary"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "kernel_ms", "converged_frac", "iters_max", "roofline_frac", "resident_per_cu", "problems_launched", "skipped_masked_frac")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_thread"]["value"])
PY
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof3
for wl in cfg2 cfg2_filtered cfg3 cfg4 lmpc cfg5 races game overtake; do
  st=50; [ $wl = cfg4 ] && st=5; [ $wl = lmpc ] && st=10; [ $wl = races ] && st=30; [ $wl = cfg5 ] && st=8; [ $wl = game ] && st=40; [ $wl = overtake ] && st=40
  rm -rf $P/$wl
  rocprofv3 --kernel-trace --stats -d $P/$wl -o $wl -- python $R/bench.py --steps $st --warmup 3 --workload $wl --no-cpu-baseline > $P/bench_$wl.json 2> $P/err_$wl.log
  db=$(find $P/$wl -name "*.db" | head -1)
  python3 $R/profiles/summarize.py $db $P/bench_$wl.json > $P/r03_${wl}_kernel_stats.txt 2>> $P/err_$wl.log
  find $P/$wl -type f ! -name "*.txt" -delete
done
ls $P | head -30
M=$R/gpurun_out/pmc3
for wl in cfg2 cfg3 cfg4 cfg5; do
  st=10; [ $wl = cfg4 ] && st=3; [ $wl = cfg5 ] && st=3
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $M/${wl}_$ctr
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $M/${wl}_$ctr -o p -- python $R/bench.py --steps $st --warmup 2 --workload $wl --no-cpu-baseline > /dev/null 2> $M/err_${wl}_$ctr.log
  done
done
cd $R && python3 profiles/summarize_pmc.py r03 pmc3 | tail -12
cp profiles/r03_pmc_hbm_traffic.txt profiles/pmc_hbm_traffic.json $O/
