cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_benchsize_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -40
