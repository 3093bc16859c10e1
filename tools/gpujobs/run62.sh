cd $GRAFT_REPO_ROOT
REPS=3 STEPS=100 bash tools/ab_bench.sh "pw:PASSION_CONV_PW=1" "lean:PASSION_CONV_PW=0"
REPS=2 STEPS=60 bash tools/ab_bench.sh "passion_pw:PASSION_CONV_PW=1" "passion_lean:PASSION_CONV_PW=0" -- --workload mmformer128_passion
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_benchsize_gpu.py tests/test_fullsize_gpu.py tests/test_rfnet_gpu.py -x -q -m gpu 2>&1 | tail -3
