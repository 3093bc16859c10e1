cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/prof_r02
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02 -o st --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_r02/bench.json 2> $R/gpurun_out/prof_r02/bench.err)
ls gpurun_out/prof_r02 | head
cp gpurun_out/prof_r02/*kernel_stats.csv gpurun_out/r02c_bench_kernel_stats.csv 2>/dev/null || find gpurun_out/prof_r02 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02c_bench_kernel_stats.csv \;
head -12 gpurun_out/r02c_bench_kernel_stats.csv | cut -c1-160
bash tools/pmc_traffic.sh gpurun_out/r02c_traffic.json > gpurun_out/pmc_traffic.log 2>&1; tail -c 1500 gpurun_out/pmc_traffic.log
rm -rf gpurun_out/prof_r02/*/*kernel_trace.csv
