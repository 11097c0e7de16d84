# round 3, GPU call K: dispatch order computed by libcrx (crx_order_longest_first_dev / crx_cbf_order_dev)
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "dispatch_order" 2>&1 | tail -3
python tools/order_probe.py 2>&1 | grep -v amdgpu.ids
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged_frac']))"; }
for d in start_barrier; do
  timeout 300 python bench.py --workload cfg4 --dispatch $d --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "cfg4 $d"
  timeout 300 python bench.py --workload cfg2 --batch 4096 --dispatch $d --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "cfg2x4096 $d"
done
for d in; do
  timeout 300 python bench.py --workload lmpc --dispatch $d --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "lmpc $d"
  for k in 1 2; do
    for wl in races game overtake; do
      st=30; [ $wl != races ] && st=60
      timeout 300 python bench.py --workload $wl --race-streams $k --dispatch $d --no-cpu-baseline --steps $st --warmup 5 2> /dev/null | line "$wl sub-batches $k $d"
    done
  done
done
