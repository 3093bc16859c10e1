cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -s 2>&1 | grep -v "^$" | grep "max |dp|\|passed\|failed\|Error" | head
