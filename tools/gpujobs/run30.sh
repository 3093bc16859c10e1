cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/prof_r02e
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02e -o st --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_r02e/bench.json 2> $R/gpurun_out/prof_r02e/bench.err)
find gpurun_out/prof_r02e -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02e_bench_kernel_stats.csv \;
rm -rf gpurun_out/prof_r02e/*/*kernel_trace.csv gpurun_out/prof_r02e/*kernel_trace.csv
tail -1 gpurun_out/prof_r02e/bench.json | cut -c1-300
