#!/bin/bash
# ONE parameterised GPU pass (replaces the per-call scripts gpu_round{2,3}_*.sh, gpu_prof*.sh, gpu_pmc*.sh of rounds 1-3):
#     gpurun --timeout T -- 'bash tools/gpu_pass.sh TAG step [step ...]'
# writes everything under gpurun_out/TAG/ (scratch; what is to be kept is copied to profiles/ after review).  Steps:
#   suite     the GPU test suite                                      -> pytest_gpu.log
#   bench     the default bench line + its full record                -> bench.json, bench_full.json
#   stats     rocprofv3 --kernel-trace --stats of every workload      -> <wl>_kernel_stats.txt (profiles/summarize.py)
#   pmc       FETCH_SIZE / WRITE_SIZE passes (separate runs, no trace domains beside --kernel-trace) -> pmc_hbm_traffic.{txt,json}
#   issue     SQ issue / wait / instruction counters of the solver kernels -> pmc_issue.txt (profiles/summarize_issue.py)
#   trace     per-phase shader cycles of one solve, on tools/ab/libcrx_trace.so (built beforehand with -DCRX_PHASE_CLOCKS) -> phase_cycles.txt
#   budget    tools/budget_experiment.sh (solver budgets vs closed-loop quality) -> budget_experiment.txt
#   lines     one line per workload (bench.py --workload ...), for A/B-ing; CRX_LIB selects another build
#   ab:NAME   the `lines` step with CRX_LIB=tools/ab/libcrx_NAME.so (tools/build_variant.sh NAME "FLAGS" beforehand, in the build container)
#   suite:NAME the GPU parity suite on tools/ab/libcrx_NAME.so
#   gen:NAME   the tests that run the GENERAL instantiations (descriptor fuzz, non-tuned horizons, 4..6 obstacles) on tools/ab/libcrx_NAME.so
#   bits:NAME  bit-for-bit comparison of the in-tree library with tools/ab/libcrx_NAME.so on the solver draws and closed loops (tools/cbf_ab.py)
#   quick, quick:NAME  the four solver workloads of `lines` only (in-tree library / tools/ab/libcrx_NAME.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
make -C oracle -s
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1: %.4g /s converged (%.4g launched)  %.4f ms/step  kernel %.4f ms  status %s  iters p50 %s p90 %s max %s' % (d['value'], d['value_launched'], d['ms_per_step'], d['roofline']['kernel_ms'], c['status_frac'], c['iters_p50'], c['iters_p90'], c['iters_max']))"; }
lines() {  # one line per workload
  timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "cfg2"
  timeout 300 python bench.py --workload cfg2 --batch 4096 --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "cfg2x4096"
  timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 100 --warmup 5 2> /dev/null | line "cfg3"
  timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | line "cfg4"
  timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | line "cfg5"
  timeout 300 python bench.py --workload lmpc --no-cpu-baseline --steps 30 --warmup 3 2> /dev/null | line "lmpc"
  timeout 300 python bench.py --workload races --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "races"
  timeout 300 python bench.py --workload game --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "game"
  timeout 300 python bench.py --workload overtake --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "overtake"
}
quick() {  # the four solver workloads only (A/B-ing kernel builds)
  timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 10 2> /dev/null | line "cfg2"
  timeout 300 python bench.py --workload cfg2 --batch 4096 --no-cpu-baseline --steps 30 --warmup 5 2> /dev/null | line "cfg2x4096"
  timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --steps 100 --warmup 5 2> /dev/null | line "cfg3"
  timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --steps 10 --warmup 2 2> /dev/null | line "cfg4"
}
for step in "$@"; do
case $step in
game:*)   # the learning-MPC workloads on tools/ab/libcrx_NAME.so ("game:" alone = the in-tree library)
  n=${step#game:}; [ -n "$n" ] && export CRX_LIB=$R/tools/ab/libcrx_$n.so; echo "== lmpc workloads ${n:-in-tree}"
  ( timeout 300 python bench.py --workload lmpc --no-cpu-baseline --steps 30 --warmup 3 2> /dev/null | line "lmpc"
    timeout 300 python bench.py --workload game --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "game"
    timeout 300 python bench.py --workload overtake --no-cpu-baseline --steps 60 --warmup 5 2> /dev/null | line "overtake" ) | tee $O/game_${n:-intree}.txt; unset CRX_LIB ;;
bits:*)  # bit-for-bit A/B of the in-tree library against tools/ab/libcrx_NAME.so (tools/cbf_ab.py: solver draws + closed loops)
  n=${step#bits:}
  python tools/cbf_ab.py intree > /dev/null 2>&1; CRX_LIB=$R/tools/ab/libcrx_$n.so python tools/cbf_ab.py $n > /dev/null 2>&1
  python tools/lmpc_ab.py intree > /dev/null 2>&1; CRX_LIB=$R/tools/ab/libcrx_$n.so python tools/lmpc_ab.py $n > /dev/null 2>&1
  ( python tools/cbf_ab.py --compare intree $n; python tools/lmpc_ab.py --compare intree $n ) | tee $O/bits_$n.txt ;;
quick) quick | tee $O/quick.txt ;;
quick:*)
  n=${step#quick:}; export CRX_LIB=$R/tools/ab/libcrx_$n.so; echo "== $n"; quick | tee $O/quick_$n.txt; unset CRX_LIB ;;
suite)
  ( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -40 ) > $O/pytest_gpu.log
  grep -h "passed\|failed" $O/pytest_gpu.log | tail -2 ;;
suite:*)
  n=${step#suite:}
  ( CRX_LIB=$R/tools/ab/libcrx_$n.so timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -30 ) > $O/pytest_$n.log
  echo "$n: $(grep -h 'passed\|failed' $O/pytest_$n.log | tail -1)" ;;
gen:*)   # the general-instantiation tests (descriptor fuzz, non-tuned horizons at batch 256, 4..6 obstacles) on tools/ab/libcrx_NAME.so
  n=${step#gen:}
  ( CRX_LIB=$R/tools/ab/libcrx_$n.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fuzz_descriptors or general_horizons or many_obstacles or no_stale_lds" 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -30 ) > $O/pytest_gen_$n.log
  echo "$n: $(grep -h 'passed\|failed' $O/pytest_gen_$n.log | tail -1)" ;;
bench)
  python bench.py --full-out $O/bench_full.json > $O/bench.json 2> $O/bench.err
  wc -c $O/bench.json; cat $O/bench.json ;;
lines) lines | tee $O/lines.txt ;;
ab:*)
  n=${step#ab:}; export CRX_LIB=$R/tools/ab/libcrx_$n.so; echo "== $n"; lines | tee $O/lines_$n.txt; unset CRX_LIB ;;
stats)
  P=$O/prof; mkdir -p $P
  ( cd /tmp && export TMPDIR=/tmp
  for wl in cfg2 cfg2_filtered cfg3 cfg4 lmpc cfg5 races game overtake; do
    st=50; [ $wl = cfg4 ] && st=5; [ $wl = lmpc ] && st=10; [ $wl = races ] && st=30; [ $wl = cfg5 ] && st=8; [ $wl = game ] && st=40; [ $wl = overtake ] && st=40
    rm -rf $P/$wl
    rocprofv3 --kernel-trace --stats -d $P/$wl -o $wl -- python $R/bench.py --steps $st --warmup 3 --workload $wl --no-cpu-baseline --full-out $P/full_$wl.json > $P/bench_$wl.json 2> $P/err_$wl.log
    db=$(find $P/$wl -name "*.db" | head -1)
    python3 $R/profiles/summarize.py $db $P/bench_$wl.json > $O/${wl}_kernel_stats.txt 2>> $P/err_$wl.log
    find $P/$wl -type f ! -name "*.txt" -delete
  done )
  ls $O | grep kernel_stats ;;
pmc)
  M=$O/pmc; mkdir -p $M
  python -c "import bench; print(bench.kernel_source_hash())" > $M/source_hash.txt
  ( cd /tmp && export TMPDIR=/tmp
  for wl in cfg2 cfg3 cfg4 cfg5; do
    st=10; [ $wl = cfg4 ] && st=3; [ $wl = cfg5 ] && st=3
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rm -rf $M/${wl}_$ctr
      rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $M/${wl}_$ctr -o p -- python $R/bench.py --steps $st --warmup 2 --workload $wl --no-cpu-baseline > /dev/null 2> $M/err_${wl}_$ctr.log
    done
  done )
  find $M -type f ! -name "*counter_collection.csv" ! -name "*.txt" ! -name "*.log" -delete
  ls $M ;;   # summarise locally: python3 profiles/summarize_pmc.py rNN $TAG/pmc
issue)
  ( cd /tmp && export TMPDIR=/tmp
  pmc() {  # tag workload batch steps extra-bench-args counters...
    tag=$1; wl=$2; b=$3; st=$4; ex=$5; shift 5
    rm -rf $O/pmc_$tag
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python $R/bench.py --steps $st --warmup 1 --workload $wl --batch $b $ex --no-cpu-baseline > /dev/null 2> $O/err_$tag.log
    find $O/pmc_$tag -type f ! -name "*counter_collection.csv" -delete
  }
  A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
  B="SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  pmc cfg4_default_1 cfg4 16384 2 "--dispatch index" $A
  pmc cfg4_default_3 cfg4 16384 2 "--dispatch index" $B
  pmc cfg2_1 cfg2 16384 3 "--dispatch index" $A
  pmc cfg2_3 cfg2 16384 3 "--dispatch index" $B
  Cc="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"    # [r5] the sweeps are unrolled 12x / 20x since round 4: does the loop body still sit in the 64 KB instruction cache two CUs share?
  pmc cfg4_default_5 cfg4 16384 2 "--dispatch index" $Cc
  pmc cfg2_5 cfg2 16384 3 "--dispatch index" $Cc
  pmc cfg3_1 cfg3 16384 3 "" $A
  pmc cfg3_3 cfg3 16384 3 "" $B )
  ls $O | grep pmc_ ;;   # summarise locally: python3 profiles/summarize_issue.py $TAG it4,it2,it3 > profiles/rNN_pmc_issue.txt
trace)   # needs tools/ab/libcrx_trace.so: tools/build_variant.sh trace "-DCRX_PHASE_CLOCKS" crx_kernels.hip crx_kernels_obs.hip crx_lmpc.hip (build container)
  ( export CRX_LIB=$R/tools/ab/libcrx_trace.so; for w in cfg2 cfg3 cfg4; do python tools/gpu_solve_trace.py $w; done; python tools/gpu_lmpc_trace.py ) 2>&1 | grep -v "amdgpu.ids\|Warning\|print(\|ret = " > $O/phase_cycles.txt
  cat $O/phase_cycles.txt ;;
budget)
  bash tools/budget_experiment.sh ${BUDGET_B:-4096} ${BUDGET_STEPS:-400} > /dev/null; cp gpurun_out/budget_experiment.txt $O/; grep "^==\|contact\|left the track\|laps completed\|lap :" $O/budget_experiment.txt ;;
*) echo "unknown step $step" ;;
esac
done
