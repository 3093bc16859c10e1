#!/bin/bash
# every launch of the LAST replayed step whose kernel name matches a pattern, in launch order, with grid and duration:  tools/trace_family.sh "REGEX" [bench args]
PAT=$1; shift
R=$PWD
rm -rf /tmp/trf; mkdir -p /tmp/trf
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/trf -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-dice --no-roofline --steps 8 --warmup 2 "$@" > /tmp/trf/bench.log 2>&1
cd $R
T=$(find /tmp/trf -name "*kernel_trace.csv" | head -1)
python3 - "$T" "$PAT" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r["Kernel_Name"]]
ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] - i > 200]
rows = rows[ends[-2] + 1: ends[-1] + 1]
t0 = int(rows[0]["Start_Timestamp"])
pat = re.compile(sys.argv[2])
tot = 0.0
for r in rows:
    if not pat.search(r["Kernel_Name"]): continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60]
    print(f"t={(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  {d:7.1f} us  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '')} q{r.get('Queue_Id', '')}  {name}")
print(f"total {tot:.1f} us")
PY
