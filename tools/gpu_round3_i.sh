# round 3, GPU call I: closed-loop workloads, sub-batches x hardware queues (GPU_MAX_HW_QUEUES), same steps as the default bench
R=$GRAFT_REPO_ROOT
cd $R
for q in 4 8 16; do
  for wl in races game overtake; do
    st=30; [ $wl != races ] && st=60
    for k in 1 2 4; do
      GPU_MAX_HW_QUEUES=$q python bench.py --workload $wl --race-streams $k --no-cpu-baseline --steps $st --warmup 5 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q $wl sub-batches $k: %.4g steps/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"
    done
  done
done
