cd $GRAFT_REPO_ROOT
BENCH_HOST=1 timeout 900 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | cut -c1-260
BENCH_HOST=1 timeout 900 python bench.py --no-cpu-baseline --no-roofline --workload mmformer128_passion 2>&1 | grep -v amdgpu.ids | cut -c1-260
BENCH_HOST=1 timeout 900 python bench.py --no-cpu-baseline --no-roofline --size 32 --steps 40 2>&1 | grep -v amdgpu.ids | cut -c1-260
timeout 600 python tools/host_profile.py 32 2>&1 | head -90
