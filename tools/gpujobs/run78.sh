cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_benchsize_gpu.py -x -q -m gpu 2>&1 | tail -3
REPS=3 STEPS=100 bash tools/ab_bench.sh "fuse_side:PASSION_FUSE_STREAM=1" "fuse_main:PASSION_FUSE_STREAM=0"
REPS=2 STEPS=60 bash tools/ab_bench.sh "p_fuse_side:PASSION_FUSE_STREAM=1" "p_fuse_main:PASSION_FUSE_STREAM=0" -- --workload mmformer128_passion
