cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "statistics" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_benchsize_gpu.py tests/test_depth_gpu.py tests/test_rfnet_gpu.py -x -q -m gpu 2>&1 | tail -3
REPS=2 bash tools/ab_bench.sh "now:PASSION_X=1"
REPS=1 bash tools/ab_bench.sh "passion:PASSION_X=1" -- --workload mmformer128_passion
