cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -5
BENCH_HOST=1 timeout 600 python bench.py --no-cpu-baseline 2>gpurun_out/r102.err | tail -1 | cut -c1-200
grep host gpurun_out/r102.err
BENCH_HOST=1 timeout 600 python bench.py --no-cpu-baseline 2>gpurun_out/r102.err | tail -1 | cut -c1-200
grep host gpurun_out/r102.err
