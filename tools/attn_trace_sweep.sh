#!/bin/bash
# kernel-time (not host-time) sweep of the fused attention over shapes and split factors: rocprofv3 --kernel-trace --stats on
# tools/attn_microbench.py, sum of the average durations of the flash_* kernels of one forward + backward
#   usage: tools/attn_trace_sweep.sh "B N" ... -- split ...
R=${GRAFT_REPO_ROOT:-$PWD}
shapes=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do shapes+=("$1"); shift; done; shift
splits=("$@"); [ ${#splits[@]} -eq 0 ] && splits=(0)
cd /tmp && export TMPDIR=/tmp
for shape in "${shapes[@]}"; do
  set -- $shape
  for sp in "${splits[@]}"; do
    D=$R/gpurun_out/attn_trace/$1_$2_$sp; rm -rf $D; mkdir -p $D
    export PASSION_FLASH_SPLIT=$sp
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/attn_microbench.py --B $1 --N $2 --iters 20 > $D/log.txt 2>&1
    python3 - $D "$1 $2 split=$sp" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
tot = 0.0; parts = []
for r in csv.DictReader(open(f)):
    if "flash_" in r["Name"]:
        a = float(r["AverageNs"]) / 1e3; tot += a
        parts.append(f'{r["Name"].split("::")[-1].split("(")[0][:22]} {a:.1f}')
print(f"B N = {sys.argv[2]}: {tot:.1f} us  [" + "; ".join(parts) + "]")
PY
  done
done
