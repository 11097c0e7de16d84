# round 3, GPU call B: which build passes the -m gpu suite?  base = lane maps hoisted (round-2 behaviour, all of this round's other
# changes), v3n = opaque lane except for the full-layout 3-obstacle instantiations, v2 = opaque Ctx copy only, default = opaque everywhere.
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r3b
mkdir -p $O
make -C oracle -s
run() {  # lib-name pytest-args...
  n=$1; shift
  L=""; [ $n != default ] && L=$R/tools/ab/libcrx_$n.so
  ( CRX_LIB=$L timeout 900 python -m pytest tests -m gpu -q "$@" 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -25 ) > $O/pytest_$n.log
  echo "== $n: $(tail -1 $O/pytest_$n.log)"
}
run base
run v3n
run v2 -x -k "overtake_step or cfg4_full or multi_agents or fuzz or draws_cbf"
run default -x -k "cfg4_full or fuzz"
grep -h "passed\|failed\|FAILED\|Error\|fault" $O/pytest_*.log | head -40
