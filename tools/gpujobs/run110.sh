cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/prof110
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof110 -o up -- python3 $R/tools/upsample_bench.py > /dev/null 2>&1)
f=$(find gpurun_out/prof110 -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "upsample" in n:
        key = (n[:60], r["Grid_Size_X"], r["Grid_Size_Y"])
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    print(k, len(v), " ".join("%.0f" % x for x in v[::max(1, len(v)//24)]))
PY
rm -rf gpurun_out/prof110
