R=$PWD
mkdir -p $R/gpurun_out/hsa
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hsa-trace --hip-trace -d $R/gpurun_out/hsa -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-dice --no-roofline --steps 4 --warmup 2 > $R/gpurun_out/hsa/bench.log 2>&1
cd $R
ls -la gpurun_out/hsa/* | head
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/hsa/**/*hip_api_trace.csv", recursive=True)
h = glob.glob("gpurun_out/hsa/**/*hsa_api_trace.csv", recursive=True)
print(f, h)
hip = list(csv.DictReader(open(f[0])))
gl = [r for r in hip if r["Function"] == "hipGraphLaunch"]
print("hipGraphLaunch calls", len(gl))
for r in gl[-3:]:
    print("  launch dur us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
last = gl[-1]
s, e, tid = int(last["Start_Timestamp"]), int(last["End_Timestamp"]), last["Thread_Id"]
hsa = [r for r in csv.DictReader(open(h[0])) if s <= int(r["Start_Timestamp"]) <= e]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in hsa:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[(r["Function"], r["Thread_Id"] == tid)][0] += 1; agg[(r["Function"], r["Thread_Id"] == tid)][1] += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{v[1]:10.1f} us n={v[0]:6d} {k}")
waits = [(int(r["Start_Timestamp"]) - s, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Function"]) for r in hsa if "wait" in r["Function"] and r["Thread_Id"] == tid]
print("long waits on the launching thread (offset us, dur us):")
for o, d, fn in waits:
    if d > 30: print(f"   {o/1e3:9.1f} {d:8.1f} {fn}")
PY
rm -rf gpurun_out/hsa
