set -x
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -40
