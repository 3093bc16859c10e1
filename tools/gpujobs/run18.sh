cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/determinism_check.py 64 passion 2>&1 | grep -v amdgpu | head -60
