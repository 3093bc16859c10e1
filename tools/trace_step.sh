#!/bin/bash
# every launch of the last step of a bench.py run, longest first (rocprofv3 --kernel-trace): usage tools/trace_step.sh OUT.txt TOP [env=..] -- [bench args]
OUT=$1; TOP=$2; shift 2
while [ $# -gt 0 ] && [ "$1" != "--" ]; do export "$1"; shift; done; [ "$1" == "--" ] && shift
R=${GRAFT_REPO_ROOT:-$PWD}
D=$R/gpurun_out/trstep; rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $D -o tr --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --prime 2 --no-cpu-baseline --no-roofline --no-dice "$@" > $D/log.txt 2>&1
cd $R
python3 tools/trace_each.py $(find $D -name "*kernel_trace.csv" | head -1) $TOP > $OUT
rm -rf $D
