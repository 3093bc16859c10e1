cd $GRAFT_REPO_ROOT
timeout 900 python tools/linear_bias_determinism.py 1000 2>&1 | grep -v amdgpu | tail -30
echo "== step"
timeout 900 python tools/determinism_check.py 64 passion train 40 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
