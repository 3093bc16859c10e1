cd $GRAFT_REPO_ROOT
timeout 600 python tools/fused_stats_bench.py 2>&1 | grep -v amdgpu
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_depth_gpu.py -x -q -m gpu 2>&1 | tail -3
REPS=2 bash tools/ab_bench.sh "fused:PASSION_FUSE_STATS=1"
timeout 900 python tools/host_profile.py plain 20 2>&1 | grep -v amdgpu | head -12
