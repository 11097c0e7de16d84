# LDS / issue counters of the solver kernel (one rocprofv3 --pmc pass per counter group; no trace domains)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_lds
cd /tmp && export TMPDIR=/tmp
for wl in cfg3 cfg5 cfg4; do
 st=10; [ $wl = cfg4 ] && st=2; [ $wl = cfg5 ] && st=3
 i=0
 for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_lds/${wl}_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lds/${wl}_$i -o p -- python $R/bench.py --steps $st --warmup 1 --workload $wl --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_lds/err_${wl}_$i.log
 done
done
find $R/gpurun_out/pmc_lds -name "*counter_collection.csv" | head -20
