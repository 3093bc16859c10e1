cd $GRAFT_REPO_ROOT
timeout 600 python tools/fused_stats_bench.py 2>&1 | grep -v amdgpu | head -3
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_depth_gpu.py tests/test_benchsize_gpu.py tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu | tail -1 | cut -c1-190
PASSION_FUSE_STATS=0 timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu | tail -1 | cut -c1-190
done
timeout 600 python bench.py --no-cpu-baseline --no-roofline --passion 2>&1 | grep -v amdgpu | tail -1 | cut -c1-190
