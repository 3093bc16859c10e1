#!/bin/bash
# per-kernel durations of a bench.py run (rocprofv3 --kernel-trace --stats): usage tools/kernel_ab.sh TAG [bench args]; writes gpurun_out/kab_TAG.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
D=$R/gpurun_out/kab_$TAG; rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $D -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-dice --no-roofline --steps 40 "$@" > $D/log.txt 2>&1
cd $R
python3 tools/kstats.py $(find $D -name "*kernel_stats.csv" | head -1) 48 45 > gpurun_out/kab_$TAG.txt
rm -rf $D
