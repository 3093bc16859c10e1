#!/bin/bash
# rocprofv3 kernel trace of the (graph-replayed) bench step -> timeline / per-kernel totals of one step.  usage: tools/trace_graph.sh TAG [bench args]
TAG=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/tr_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tr_$TAG -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-dice --no-roofline --steps 12 --warmup 2 "$@" > $R/gpurun_out/tr_$TAG/bench.log 2>&1
cd $R
T=$(find gpurun_out/tr_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $T > gpurun_out/${TAG}_timeline.txt 2>&1
python3 tools/trace_last.py $T 70 > gpurun_out/${TAG}_trace_last.txt 2>&1
python3 tools/trace_bins.py $T > gpurun_out/${TAG}_trace_bins.txt 2>&1
python3 tools/trace_idle.py $T 12 > gpurun_out/${TAG}_trace_idle.txt 2>&1
python3 tools/trace_conc.py $T > gpurun_out/${TAG}_trace_conc.txt 2>&1
python3 tools/trace_seq.py $T > gpurun_out/${TAG}_trace_seq.txt 2>&1
cp $(find gpurun_out/tr_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
tail -2 gpurun_out/tr_$TAG/bench.log | cut -c1-400
rm -rf gpurun_out/tr_$TAG
