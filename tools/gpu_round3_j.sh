# round 3, GPU call J: longest-first dispatch (crx_*_solve_ordered_dev) -- bit-identity test, then index order vs longest first
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "dispatch_order or masked_launches or cfg4_full_size" 2>&1 | tail -3
for wl in cfg4 lmpc races game overtake; do
  st=30; [ $wl = game -o $wl = overtake ] && st=60
  for d in index longest_first index longest_first; do
    timeout 300 python bench.py --workload $wl --dispatch $d --no-cpu-baseline --steps $st --warmup 5 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl $d: %.4g /s  %.4f ms/step  kernel %.4f ms  conv %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged_frac']))"
  done
done
