#!/bin/bash
# HIP runtime API calls of a training step (host side): rocprofv3 --hip-trace --stats over tools/host_profile.py at 32^3
R=${GRAFT_REPO_ROOT:-$PWD}
D=$R/gpurun_out/hipapi; rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp S=32
rocprofv3 --hip-trace --stats -d $D -o h --output-format csv -- python3 $R/tools/host_profile.py plain 20 > $D/log.txt 2>&1
cd $R
f=$(find $D -name "*hip_api_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 5 + 20 + 20            # warm-up + timed + profiled loops of host_profile.py
for r in rows[:22]:
    print(f'{r["Name"][:44]:44s} calls/step {int(r["Calls"]) / steps:8.1f}  total/step {float(r["TotalDurationNs"]) / 1e3 / steps:9.1f} us  avg {float(r["AverageNs"]) / 1e3:7.2f} us')
PY
rm -rf $D
