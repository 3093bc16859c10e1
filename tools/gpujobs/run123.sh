cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "zero_padded or conv_block or kernel_paths or grouped" 2>&1 | tail -6
