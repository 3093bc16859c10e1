cd $GRAFT_REPO_ROOT
R=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "upsample" 2>&1 | tail -2
mkdir -p gpurun_out/prof97
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof97 -o st --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 20 --workload mmformer128_passion > $R/gpurun_out/prof97/bench.json 2> $R/gpurun_out/prof97/bench.err)
find gpurun_out/prof97 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r97_kernel_stats.csv \;
rm -rf gpurun_out/prof97
python tools/kstats.py gpurun_out/r97_kernel_stats.csv 35 | grep -i "upsample\|kernel time"
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --workload mmformer128_passion 2>/dev/null | tail -1 | cut -c1-160; done
