cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_rfnet_gpu.py tests/test_benchsize_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-roofline --workload rfnet128_passion 2>/dev/null | tail -1 | cut -c1-200
