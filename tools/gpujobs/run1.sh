set -x
cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
timeout 1500 python -m pytest tests/test_benchsize_gpu.py -x -q -s -m gpu > gpurun_out/r02_benchsize.log 2>&1; echo "benchsize rc $?"
tail -30 gpurun_out/r02_benchsize.log
timeout 900 python bench.py > gpurun_out/r02a_bench_plain.json 2> gpurun_out/r02a_bench_plain.err; echo "bench rc $?"
cat gpurun_out/r02a_bench_plain.json
timeout 600 python bench.py --workload mmformer128_passion --no-cpu-baseline > gpurun_out/r02a_bench_passion.json 2> gpurun_out/r02a_bench_passion.err; echo "bench passion rc $?"
cat gpurun_out/r02a_bench_passion.json
PASSION_DEVICE_INDEX=0 PASSION_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r02a_bench_2rank.json 2> gpurun_out/r02a_bench_2rank.err; echo "bench 2rank rc $?"
cat gpurun_out/r02a_bench_2rank.json; tail -3 gpurun_out/r02a_bench_2rank.err
timeout 600 bash tools/pmc_attention.sh gpurun_out/r02a_attention_pmc_old.json > gpurun_out/pmc_attn_old.log 2>&1; echo "pmc rc $?"
tail -40 gpurun_out/pmc_attn_old.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_benchsize_gpu.py > gpurun_out/r02_gputests.log 2>&1; echo "gpu tests rc $?"
tail -5 gpurun_out/r02_gputests.log
