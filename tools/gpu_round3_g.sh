# round 3, GPU call G: closed-loop workloads as concurrent sub-batches on HIP streams (crx.montecarlo.Concurrent) -- test + bench sweep
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3g
mkdir -p $O
make -C oracle -s
( timeout 900 python -m pytest tests/test_gpu_closed_loop.py -m gpu -q -k "concurrent or fused" 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -15 ) > $O/pytest.log
grep -h "passed\|failed" $O/pytest.log | tail -2
for wl in races game overtake; do
  for k in 1 2 3 4; do
    python bench.py --workload $wl --race-streams $k --no-cpu-baseline --steps 60 --warmup 5 2> $O/err_${wl}_$k.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl streams $k: %.4g steps/s  %.4f ms/step  kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" || tail -3 $O/err_${wl}_$k.log
  done
done
