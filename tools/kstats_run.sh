#!/bin/bash
# rocprofv3 kernel statistics of one bench workload:  tools/kstats_run.sh WORKLOAD TAG   ->  gpurun_out/TAG_kernel_stats.csv (+ top 25 on stdout)
cd /tmp && export TMPDIR=/tmp
W=$1; TAG=$2
rm -rf /tmp/ks_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/ks_$TAG -o ks --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-dice > /tmp/ks_$TAG.log 2>&1
f=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { tail -5 /tmp/ks_$TAG.log; exit 1; }; cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{len(rows)} kernels, {tot / 1e6 / 24:.2f} ms of kernel time per step (24 steps incl. warm-up)")
for r in rows[:25]:
    print(f'{float(r["TotalDurationNs"]) / 24e3:9.1f} us/step {int(r["Calls"]) / 24:7.1f}x {float(r["Percentage"]):5.1f}%  {r["Name"][:110]}')
PY
