cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_depth_gpu.py -x -q -s -m gpu 2>&1 | tail -40 | cut -c1-250
