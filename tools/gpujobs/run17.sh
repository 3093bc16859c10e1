cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/prof_r02p
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02p -o st --output-format csv -- python3 $R/bench.py --no-cpu-baseline --workload mmformer128_passion --steps 20 > $R/gpurun_out/prof_r02p/bench.json 2> $R/gpurun_out/prof_r02p/bench.err)
find gpurun_out/prof_r02p -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02c_bench_passion_kernel_stats.csv \;
rm -rf gpurun_out/prof_r02p
for pl in 10 12 16; do echo "planes $pl"; PASSION_MARCH_PLANES=$pl timeout 200 python tools/conv_microbench.py bwd 8 8 128 2>&1 | tail -3 | head -2; done
