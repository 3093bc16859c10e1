cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_parallel_gpu.py tests/test_depth_gpu.py -x -q -m gpu 2>&1 | tail -4
REPS=3 STEPS=100 bash tools/ab_bench.sh "defer:PASSION_DEFER_FOLD=1" "fold_each:PASSION_DEFER_FOLD=0"
