"""summarise a rocprofv3 --kernel-trace CSV: per kernel name (shortened) and grid size: calls, avg/total time"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    name = r["Kernel_Name"]
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)[:70]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1
    agg[key][1] += d
    tot += d
print(f"total {tot/1e3:.2f} ms over {len(rows)} dispatches")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{t/1e3:9.3f} ms  n={n:5d} avg {t/n:9.1f} us  grid=({k[1]},{k[2]})  {k[0]}")
