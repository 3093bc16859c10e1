cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_depth_gpu.py tests/test_rfnet_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/linear_bias_determinism.py 500 2>&1 | grep -v amdgpu | grep "concurrent=True" | grep -v "torch sum"
REPS=2 STEPS=100 bash tools/ab_bench.sh "now:X=1"
REPS=2 STEPS=60 bash tools/ab_bench.sh "passion:X=1" -- --workload mmformer128_passion
