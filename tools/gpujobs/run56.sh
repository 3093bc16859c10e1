cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_parallel_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/host_profile.py plain 20 2>&1 | grep -v amdgpu | head -3
REPS=3 STEPS=100 bash tools/ab_bench.sh "now:X=1"
