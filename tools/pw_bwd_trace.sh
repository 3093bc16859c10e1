#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for cfg in "16 4 128" "8 4 128" "32 4 64"; do
  D=$R/gpurun_out/pwtr; rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --stats -d $D -o p --output-format csv -- python3 $R/tools/pw_bwd_trace.py $cfg > $D/log.txt 2>&1
  tail -1 $D/log.txt
  python3 $R/tools/kstats.py $(find $D -name "*kernel_stats.csv" | head -1) 12 12 | grep -v "elementwise\|fill\|Fill\|distribution\|copy" | head -9
done
rm -rf $R/gpurun_out/pwtr
