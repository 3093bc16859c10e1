cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -2 > gpurun_out/bench_r02d_plain.json
timeout 600 python bench.py --no-cpu-baseline --passion 2>&1 | grep -v amdgpu | tail -2 > gpurun_out/bench_r02d_passion.json
tail -c 3000 gpurun_out/bench_r02d_plain.json
