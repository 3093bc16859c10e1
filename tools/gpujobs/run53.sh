cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -s 2>&1 | grep -v "^$" | grep "sample 1 alone\|passed\|failed"
PASSION_GROUPED_ENC=0 PASSION_FUSE_STATS=0 timeout 1200 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -s 2>&1 | grep -v "^$" | grep "sample 1 alone\|passed\|failed"
