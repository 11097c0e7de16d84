# round 3, GPU call U: issue / wait / instruction counters of the solver kernels of the FINAL build (rocprofv3 --pmc, csv under
# gpurun_out/r3u/; profiles/summarize_issue.py r3u -> profiles/r03_pmc_issue_final.txt)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
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
pmc cfg3_1 cfg3 16384 3 "" $A
pmc cfg3_3 cfg3 16384 3 "" $B
pmc cfg3off_1 cfg3 16384 3 "--no-reach-screen" $A
pmc cfg3off_3 cfg3 16384 3 "--no-reach-screen" $B
ls $O
grep -l -i "error\|invalid" $O/err_*.log | head
