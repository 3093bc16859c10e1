"""per-kernel totals over the LAST training step of a rocprofv3 --kernel-trace CSV (steps are delimited by the
multi-tensor AdamW kernels): python tools/trace_last.py trace.csv [top]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "adamw_multi_kernel" in r["Kernel_Name"]]
# group optimizer launches into steps: a gap of > 200 dispatches separates steps
ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] - i > 200]
if len(ends) >= 2:
    rows = rows[ends[-2] + 1: ends[-1] + 1]
agg = collections.defaultdict(lambda: [0, 0.0]); tot = 0.0
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"void ", "", name)[:64]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[name][0] += 1; agg[name][1] += d; tot += d
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
print(f"last step: {len(rows)} dispatches, kernel time {tot/1e3:.2f} ms, wall span {span:.2f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t/1e3:9.3f} ms n={c:5d} avg {t/c:8.1f} us  {k}")
