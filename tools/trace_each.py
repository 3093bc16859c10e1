"""individual launches of the LAST training step of a rocprofv3 --kernel-trace CSV, longest first, with grid size:
python tools/trace_each.py trace.csv [top]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "adamw_multi_kernel" in r["Kernel_Name"]]
ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] - i > 200]
if len(ends) >= 2:
    rows = rows[ends[-2] + 1: ends[-1] + 1]
out = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"void ", "", name)[:56]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.append((d, name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", "")))
tot = sum(o[0] for o in out)
print(f"{len(out)} dispatches, {tot/1e3:.2f} ms kernel time")
acc = 0.0
for d, n, gx, gy, wg in sorted(out, key=lambda o: -o[0])[:top]:
    acc += d
    print(f"{d:8.1f} us  cum {acc/1e3:6.2f} ms  grid {gx}x{gy} wg {wg}  {n}")
