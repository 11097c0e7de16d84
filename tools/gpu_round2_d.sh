R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2d
mkdir -p $O
rm -f $R/gpurun_out/parity_report.jsonl
make -C $R/oracle -s
cd $R
CRX_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "solver time\|^overtaking\|local planner\|lap completed\|solver fail" | tail -150 > $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -25 $O/pytest.log; cut -c1-400 $O/bench_default.json
