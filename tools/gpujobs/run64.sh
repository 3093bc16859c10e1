cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_benchsize_gpu.py tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -5
REPS=3 STEPS=100 bash tools/ab_bench.sh "gintra:PASSION_GROUPED_INTRA=1" "streams:PASSION_GROUPED_INTRA=0"
timeout 900 python tools/host_profile.py plain 20 2>&1 | grep -v amdgpu | head -3
