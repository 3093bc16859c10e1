cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_benchsize_gpu.py tests/test_fullsize_gpu.py tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -3
REPS=2 STEPS=100 bash tools/ab_bench.sh "now:X=1"
REPS=2 STEPS=60 bash tools/ab_bench.sh "passion:X=1" -- --workload mmformer128_passion
