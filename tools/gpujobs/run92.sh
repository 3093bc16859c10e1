cd $GRAFT_REPO_ROOT
R=$PWD
timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | tail -1 | cut -c1-160
timeout 1200 python bench.py 2> gpurun_out/r02n_bench.err | tail -1 > gpurun_out/r02n_bench_mmformer128.json
timeout 900 python bench.py --no-cpu-baseline --workload mmformer128_passion --masks split 2>/dev/null | tail -1 > gpurun_out/r02n_bench_mmformer128_passion_split.json
timeout 900 python bench.py --no-cpu-baseline --workload rfnet128_passion 2>/dev/null | tail -1 > gpurun_out/r02n_bench_rfnet128_passion.json
for wl in mmformer128 mmformer128_passion; do
mkdir -p gpurun_out/prof_$wl
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$wl -o st --output-format csv -- python3 $R/bench.py --no-cpu-baseline --workload $wl > $R/gpurun_out/prof_$wl/bench.json 2> $R/gpurun_out/prof_$wl/bench.err)
find gpurun_out/prof_$wl -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02n_${wl}_kernel_stats.csv \;
rm -rf gpurun_out/prof_$wl
done
bash tools/pmc_traffic.sh gpurun_out/r02n_traffic.json > gpurun_out/pmc_traffic.log 2>&1; tail -c 300 gpurun_out/pmc_traffic.log
for f in gpurun_out/r02n_bench_*.json; do echo $f; cut -c1-230 $f; done
