cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gputests2.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r02_gputests2.log | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r02b_bench_plain.json 2> gpurun_out/r02b_bench_plain.err; echo "bench rc $?"
cat gpurun_out/r02b_bench_plain.json; tail -3 gpurun_out/r02b_bench_plain.err
timeout 600 python bench.py --workload mmformer128_passion --no-cpu-baseline > gpurun_out/r02b_bench_passion.json 2> gpurun_out/r02b_bench_passion.err; echo "bench passion rc $?"
cat gpurun_out/r02b_bench_passion.json
