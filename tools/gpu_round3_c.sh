# round 3, GPU call C: the suite on the adopted build; what saturates a CU at 3-4 resident <3,20> waves (PMC: issue / wait /
# instruction-cache counters), slim (4 per CU) against the round-2-like full layout (3 per CU).
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3c
mkdir -p $O
make -C oracle -s
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -25 ) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_INSTS_[A-Z_]*" $O/counters.txt | sort -u | tr '\n' ' ' > $O/counter_names.txt
cat $O/counter_names.txt; echo
pmc() {  # tag lib workload batch steps counters...
  tag=$1; lib=$2; wl=$3; b=$4; st=$5; shift 5
  L=""; [ $lib != default ] && L=$R/tools/ab/libcrx_$lib.so
  rm -rf $O/pmc_$tag
  CRX_LIB=$L rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python $R/bench.py --steps $st --warmup 1 --workload $wl --batch $b --no-cpu-baseline > /dev/null 2> $O/err_$tag.log
  python $R/tools/pmc_sum.py $O/pmc_$tag $((b*64)) >> $O/pmc_summary.txt 2>&1
  find $O/pmc_$tag -type f ! -name "*counter_collection.csv" -delete
}
: > $O/pmc_summary.txt
for v in default r2like; do
  echo "### cfg4 $v" >> $O/pmc_summary.txt
  pmc cfg4_${v}_1 $v cfg4 16384 2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
  pmc cfg4_${v}_2 $v cfg4 16384 2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
  pmc cfg4_${v}_3 $v cfg4 16384 2 SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
done
echo "### cfg2 batch 16384 default" >> $O/pmc_summary.txt
pmc cfg2_1 default cfg2 16384 3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pmc cfg2_2 default cfg2 16384 3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
echo "### cfg3 batch 16384 (65536 QPs) default" >> $O/pmc_summary.txt
pmc cfg3_1 default cfg3 16384 3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pmc cfg3_2 default cfg3 16384 3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
cat $O/pmc_summary.txt
grep -l -i "error\|invalid" $O/err_*.log | head
