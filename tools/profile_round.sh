#!/bin/bash
# end-of-round evidence for profiles/: bench lines of the three workloads, rocprofv3 --kernel-trace --stats summaries of the same
# commands, FETCH/WRITE counter traffic.   usage (on the GPU box, from the repo root): bash tools/profile_round.sh r03
TAG=$1
R=$PWD
mkdir -p $R/gpurun_out/prof_$TAG
python3 bench.py > $R/gpurun_out/${TAG}_bench_mmformer128.json 2> $R/gpurun_out/${TAG}_bench_mmformer128.err
python3 bench.py --workload mmformer128_passion --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_mmformer128_passion_split.json 2>/dev/null
python3 bench.py --workload rfnet128_passion --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_rfnet128_passion.json 2>/dev/null
python3 bench.py --workload m2ftrans128 --no-cpu-baseline --no-dice > $R/gpurun_out/${TAG}_bench_m2ftrans128.json 2>/dev/null
python3 bench.py --workload m2ftrans128_passion --no-cpu-baseline --no-dice > $R/gpurun_out/${TAG}_bench_m2ftrans128_passion.json 2>/dev/null
python3 bench.py --workload mmformer_depth --no-cpu-baseline --no-dice > $R/gpurun_out/${TAG}_bench_depth_n1.json 2>/dev/null
python3 bench.py --size 80 --no-cpu-baseline --no-dice --no-roofline > $R/gpurun_out/${TAG}_bench_mmformer80.json 2>/dev/null
python3 tools/launch_table.py 128 300 2>/dev/null > $R/gpurun_out/${TAG}_launch_table.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG/plain -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-dice > $R/gpurun_out/prof_$TAG/plain.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG/passion -o p --output-format csv -- python3 $R/bench.py --workload mmformer128_passion --no-cpu-baseline --no-dice > $R/gpurun_out/prof_$TAG/passion.log 2>&1
cd $R
cp $(find gpurun_out/prof_$TAG/plain -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
cp $(find gpurun_out/prof_$TAG/passion -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_passion_kernel_stats.csv
python3 tools/trace_idle.py $(find gpurun_out/prof_$TAG/plain -name "*kernel_trace.csv" | head -1) 3 | head -2 > gpurun_out/${TAG}_trace_idle.txt
rm -rf gpurun_out/prof_$TAG
bash tools/pmc_traffic.sh gpurun_out/${TAG}_traffic.json > /dev/null 2>&1
rm -rf gpurun_out/pmc_traffic
tail -c 600 gpurun_out/${TAG}_bench_mmformer128.json; echo; cat gpurun_out/${TAG}_trace_idle.txt
