#!/bin/bash
# end-of-round evidence for profiles/: bench lines of every workload (HIP-graph replay, the default; eager and the N > 1 code path at world
# size 1 beside it), rocprofv3 --kernel-trace --stats summaries of the same commands, per-step tables, counter traffic per workload,
# MFMA busy of the attention kernels.   usage (on the GPU box, from the repo root): bash tools/profile_round.sh r04
TAG=$1
R=$PWD
O=$R/gpurun_out
mkdir -p $O/prof_$TAG
python3 bench.py > $O/${TAG}_bench_mmformer128.json 2> $O/${TAG}_bench_mmformer128.err
BENCH_HOST=1 python3 bench.py --launch eager --no-cpu-baseline --no-dice > $O/${TAG}_bench_mmformer128_eager.json 2> $O/${TAG}_bench_mmformer128_eager.err
BENCH_HOST=1 python3 bench.py --ddp-mode --no-cpu-baseline --no-dice --no-roofline > $O/${TAG}_bench_mmformer128_ddpmode.json 2> $O/${TAG}_bench_mmformer128_ddpmode.err
python3 bench.py --workload mmformer128_passion --no-cpu-baseline > $O/${TAG}_bench_mmformer128_passion_split.json 2>/dev/null
python3 bench.py --workload rfnet128_passion --no-cpu-baseline > $O/${TAG}_bench_rfnet128_passion.json 2>/dev/null
python3 bench.py --workload m2ftrans128 --no-cpu-baseline --no-dice > $O/${TAG}_bench_m2ftrans128.json 2>/dev/null
python3 bench.py --workload m2ftrans128_passion --no-cpu-baseline --no-dice > $O/${TAG}_bench_m2ftrans128_passion.json 2>/dev/null
BENCH_HOST=1 python3 bench.py --workload mmformer_depth --no-cpu-baseline --no-dice > $O/${TAG}_bench_depth_n1.json 2> $O/${TAG}_bench_depth_n1.err
BENCH_HOST=1 python3 bench.py --workload mmformer_depth --launch eager --no-cpu-baseline --no-dice > $O/${TAG}_bench_depth_n1_eager.json 2> $O/${TAG}_bench_depth_n1_eager.err
BENCH_HOST=1 python3 bench.py --workload mmformer_depth --ddp-mode --no-cpu-baseline --no-dice > $O/${TAG}_bench_depth_n1_loopback.json 2> $O/${TAG}_bench_depth_n1_loopback.err
BENCH_HOST=1 python3 bench.py --ddp-mode --graph-collectives --no-cpu-baseline --no-dice --no-roofline > $O/${TAG}_bench_mmformer128_ddpmode_fullgraph.json 2> $O/${TAG}_bench_mmformer128_ddpmode_fullgraph.err
python3 bench.py --workload mmformer128_passion --ddp-mode --no-cpu-baseline --no-dice --no-roofline > $O/${TAG}_bench_mmformer128_passion_ddpmode_eager.json 2>/dev/null
python3 bench.py --workload mmformer128_passion --ddp-mode --graph-collectives --no-cpu-baseline --no-dice --no-roofline > $O/${TAG}_bench_mmformer128_passion_ddpmode_fullgraph.json 2>/dev/null
BENCH_HOST=1 python3 bench.py --size 80 --no-cpu-baseline --no-dice --no-roofline > $O/${TAG}_bench_mmformer80.json 2> $O/${TAG}_bench_mmformer80.err
BENCH_HOST=1 python3 bench.py --size 80 --launch eager --no-cpu-baseline --no-dice --no-roofline > $O/${TAG}_bench_mmformer80_eager.json 2> $O/${TAG}_bench_mmformer80_eager.err
python3 tools/launch_table.py 128 300 2>/dev/null > $O/${TAG}_launch_table.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG/plain -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-dice > $O/prof_$TAG/plain.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG/passion -o p --output-format csv -- python3 $R/bench.py --workload mmformer128_passion --no-cpu-baseline --no-dice > $O/prof_$TAG/passion.log 2>&1
cd $R
cp $(find gpurun_out/prof_$TAG/plain -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
cp $(find gpurun_out/prof_$TAG/passion -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_passion_kernel_stats.csv
T=$(find gpurun_out/prof_$TAG/plain -name "*kernel_trace.csv" | head -1)
python3 tools/trace_idle.py $T 3 | head -2 > gpurun_out/${TAG}_trace_idle.txt
python3 tools/trace_conc.py $T 25 > gpurun_out/${TAG}_trace_conc.txt
python3 tools/trace_last.py $T 60 > gpurun_out/${TAG}_trace_last.txt
rm -rf gpurun_out/prof_$TAG
timeout 900 bash tools/pmc_traffic.sh gpurun_out/${TAG}_traffic_mmformer128.json > /dev/null 2>&1
timeout 900 bash tools/pmc_traffic.sh gpurun_out/${TAG}_traffic_mmformer128_passion.json --workload mmformer128_passion > /dev/null 2>&1
timeout 900 bash tools/pmc_traffic.sh gpurun_out/${TAG}_traffic_rfnet128_passion.json --workload rfnet128_passion > /dev/null 2>&1
rm -rf gpurun_out/pmc_traffic
timeout 600 bash tools/pmc_attention_bench.sh gpurun_out/${TAG}_attention_pmc_bench_plain.json > /dev/null 2>&1
timeout 600 bash tools/pmc_attention_bench.sh gpurun_out/${TAG}_attention_pmc_bench_passion.json --workload mmformer128_passion > /dev/null 2>&1
rm -rf gpurun_out/pmc_attn_bench
# the attention kernels alone on a shape that fills the machine (B = 16, N = 2048: 8192 query tiles for 1024 SIMDs), with and without dropout
timeout 600 bash tools/pmc_attention.sh gpurun_out/${TAG}_attention_pmc_saturating.json --B 16 --N 2048 > /dev/null 2>&1
timeout 600 bash tools/pmc_attention.sh gpurun_out/${TAG}_attention_pmc_saturating_nodrop.json --B 16 --N 2048 --drop 0 > /dev/null 2>&1
rm -rf gpurun_out/pmc_attn
timeout 600 bash tools/kstats_run.sh rfnet128_passion ${TAG}_rfnet > /dev/null 2>&1
PASSION_BENCH_GRAPH=0 timeout 600 bash tools/kstats_run.sh mmformer_depth ${TAG}_depth > /dev/null 2>&1
timeout 900 python3 tools/rccl_capture_probe.py micro_ar micro_ara micro_ag micro_agt micro_p2p depth depthc_g ddp ddpp 2>&1 | grep "^\[" > gpurun_out/${TAG}_rccl_capture_probe.txt
for f in mmformer128 mmformer128_eager mmformer128_ddpmode mmformer128_ddpmode_fullgraph mmformer128_passion_split mmformer128_passion_ddpmode_eager mmformer128_passion_ddpmode_fullgraph rfnet128_passion m2ftrans128 m2ftrans128_passion depth_n1 depth_n1_eager depth_n1_loopback mmformer80 mmformer80_eager; do
  echo "$f: $(python3 -c "import json; d=json.loads(open('gpurun_out/${TAG}_bench_$f.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)"
done
grep -h "host enqueue" gpurun_out/${TAG}_bench_*.err
cat gpurun_out/${TAG}_trace_idle.txt; head -8 gpurun_out/${TAG}_trace_conc.txt; head -4 gpurun_out/${TAG}_launch_table.txt
