# first GPU pass of round 2: test suite (full report), disagreement dump, default bench line, PMC calibration
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
make -C $R/oracle -s
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_closed_loop.py 2>&1 | grep -v "solver time\|^overtaking\|local planner\|lap completed" | tail -40 > $O/pytest_parity.log
timeout 900 python -m pytest tests/test_gpu_closed_loop.py -m gpu -q 2>&1 | grep -v "solver time\|^overtaking\|local planner\|lap completed\|solver fail" | tail -30 > $O/pytest_closed_loop.log
timeout 600 python tools/parity_diff.py > $O/parity_diff.json 2> $O/parity_diff.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/gpu_calib.sh > $O/calib.log 2>&1
tail -5 $O/pytest_parity.log; tail -5 $O/pytest_closed_loop.log; cut -c1-600 $O/bench_default.json
