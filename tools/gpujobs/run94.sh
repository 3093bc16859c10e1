cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | tail -1 | cut -c1-160
timeout 1200 python bench.py 2> gpurun_out/r02m_bench.err | tail -1 > gpurun_out/r02m_bench_mmformer128.json
cut -c1-200 gpurun_out/r02m_bench_mmformer128.json
