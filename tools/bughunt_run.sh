#!/bin/bash
# GPU side of tools/bughunt_build.sh: trial 9 of the descriptor fuzz (<2,24,6,0>, restore_iters = -1) on every variant of the failing source state;
# a variant FAILS when its iteration counts / statuses leave the oracle's.  Output: gpurun_out/TAG/bughunt.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out/${1:-bughunt}; mkdir -p $O
make -C oracle -s
for so in tools/ab/libcrx_bug_${2:-*}.so; do
  n=$(basename $so .so); n=${n#libcrx_bug_}
  CRX_LIB=$R/$so timeout 120 python tools/fuzz_ab.py 9 bug_$n > /dev/null 2>&1
  python - "$n" <<'PY'
import sys, numpy as np
n = sys.argv[1]
try:
    a = np.load("gpurun_out/fuzz_ab_9_bug_%s.npz" % n)
    bad = np.nonzero((a["iters"] != a["o_iters"]) | (a["status"] != a["o_status"]))[0]
    print("%-10s %s  problems off the oracle: %s  iters %s vs oracle %s" % (n, "PASS" if len(bad) == 0 else "FAIL", bad.tolist(), a["iters"][bad].tolist(), a["o_iters"][bad].tolist()))
except Exception as e:
    print("%-10s no result (%s)" % (n, e))
PY
done | tee $O/bughunt.txt
