# GPU pass: full test suite WITHOUT -x and with the disagreement report, traces of the planner iteration flips, bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
rm -f $R/gpurun_out/parity_report.jsonl
make -C $R/oracle -s
cd $R
CRX_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "solver time\|^overtaking\|local planner\|lap completed\|solver fail" | tail -150 > $O/pytest.log
for i in 133 171; do timeout 120 python tools/parity_trace.py cfg3 12 $i > $O/trace_cfg3_12_$i.log 2>&1; done
timeout 120 python tools/parity_trace.py cfg3 20 182 > $O/trace_cfg3_20_182.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -25 $O/pytest.log; cut -c1-400 $O/bench_default.json
