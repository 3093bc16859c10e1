cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_benchsize_gpu.py tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -3
REPS=3 STEPS=100 bash tools/ab_bench.sh "loss_side:PASSION_LOSS_STREAMS=1" "loss_main:PASSION_LOSS_STREAMS=0"
REPS=2 STEPS=60 bash tools/ab_bench.sh "p_loss_side:PASSION_LOSS_STREAMS=1" "p_loss_main:PASSION_LOSS_STREAMS=0" -- --workload mmformer128_passion
