cd $GRAFT_REPO_ROOT
timeout 600 python tools/fused_stats_bench.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "statistics or colsum or linear_fn" 2>&1 | tail -5
timeout 900 python tools/linear_bias_determinism.py 300 2>&1 | grep -v amdgpu | grep -v "torch sum" | tail -12
