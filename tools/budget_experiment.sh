#!/bin/bash
# whose collisions are those?  The batched racing game with traffic (tools/multi_tests.py: the reference's
# --multi-tests experiment), same seeds, process noise OFF, under
#   (a) the product's default solver budgets (max_iter 200, restoration budget 25, reachability screens, crash path),
#   (b) IPOPT-like budgets: max_iter 3000 (IPOPT's own default), restoration budget 3000 (never the reason to stop), no screens
#       (every QP / first attempt goes through the interior-point iteration),
#   (c) libcrx 0.1.x's behaviour on crash states (slack_start = 0), default budgets.
# If contact / off-track shares move by more than a few points between (a) and (b), the early exits change the controller.
# usage (GPU box): bash tools/budget_experiment.sh [B=4096] [steps=400]   -> gpurun_out/budget_experiment.txt
B=${1:-4096}; S=${2:-400}
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
{
echo "== (a) product defaults"; python tools/multi_tests.py $B $S 3 none
echo "== (b) IPOPT-like budgets"; CRX_OPTS="max_iter=3000,restore_iters=3000,reach_screen=0" python tools/multi_tests.py $B $S 3 none
echo "== (c) slack_start = 0 (libcrx 0.1.x)"; CRX_OPTS="slack_start=0" python tools/multi_tests.py $B $S 3 none
echo "== (a') product defaults, process noise ON (seed 1)"; python tools/multi_tests.py $B $S 3 1
echo "== (b') IPOPT-like budgets, process noise ON (seed 1)"; CRX_OPTS="max_iter=3000,restore_iters=3000,reach_screen=0" python tools/multi_tests.py $B $S 3 1
} 2>&1 | tee gpurun_out/budget_experiment.txt
