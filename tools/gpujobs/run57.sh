cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_parallel_gpu.py tests/test_rfnet_gpu.py -x -q -m gpu 2>&1 | tail -3
