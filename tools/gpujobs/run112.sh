cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_benchsize_gpu.py -x -q -m gpu -k "config3_rfnet96" 2>&1 | grep -v Warn | tail -40
