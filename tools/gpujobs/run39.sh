cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_benchsize_gpu.py tests/test_depth_gpu.py tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -4
REPS=2 bash tools/ab_bench.sh "gemm1x1:PASSION_GEMM_1X1=1" "conv1x1:PASSION_GEMM_1X1=0"
