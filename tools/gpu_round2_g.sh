R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
rm -f $R/gpurun_out/parity_report.jsonl
make -C $R/oracle -s
cd $R
CRX_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "solver time\|^overtaking\|local planner\|lap completed\|solver fail\|the non-converged solution" > $O/pytest_full.log
grep -n "^E   \|^tests/test_gpu.*py:[0-9]*: \|^____\|^FAILED\|passed\|failed" $O/pytest_full.log | cut -c1-900 > $O/pytest.log
tail -40 $O/pytest.log
