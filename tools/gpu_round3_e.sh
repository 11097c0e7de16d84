# round 3, GPU call E: two-stream overlap of the racing-game branches + parallel commit kernel (closed-loop tests, bench), the
# Monte-Carlo experiment with process noise, the winners' all-gather through RCCL at world size 1 (torch and C ABI).
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3e
mkdir -p $O
make -C oracle -s
( timeout 1200 python -m pytest tests/test_gpu_closed_loop.py -m gpu -q 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -25 ) > $O/pytest_closed_loop.log
grep -h "passed\|failed" $O/pytest_closed_loop.log | tail -2
for wl in overtake game; do
  python bench.py --workload $wl --no-cpu-baseline --steps 60 --warmup 5 2> $O/bench_$wl.err > $O/bench_$wl.json
  python -c "import json; d=json.loads(open('$O/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('scene_overflow_races'), d['config']['status_frac'])"
done
python tools/multi_tests.py 4096 400 3 1 > $O/multi_tests_noise.txt 2>&1; tail -12 $O/multi_tests_noise.txt
python tools/multi_tests.py 4096 400 3 none > $O/multi_tests_zero_noise.txt 2>&1; tail -12 $O/multi_tests_zero_noise.txt
for c in torch crx; do
  python bench.py --workload cfg5 --force-collective --collective $c --no-cpu-baseline --steps 20 --warmup 3 2> $O/cfg5_collective_$c.err > $O/cfg5_collective_$c.json
  python -c "import json; d=json.loads(open('$O/cfg5_collective_$c.json').read().strip().splitlines()[-1]); print('cfg5 force-collective $c', d['value'], d['ms_per_step'], d.get('allgather_ms'), d.get('world_size'))" || tail -5 $O/cfg5_collective_$c.err
done
