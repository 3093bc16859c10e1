"""kernel time of the LAST training step binned by workgroup count (can a launch fill 256 CUs on its own?)"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "adamw_multi_kernel" in r["Kernel_Name"]]
ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] - i > 200]
if len(ends) >= 2:
    rows = rows[ends[-2] + 1: ends[-1] + 1]
bins = collections.OrderedDict((k, [0, 0.0]) for k in ("<32 WGs", "32-255", "256-1023", ">=1024"))
byname = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = 1
    for ax in "XYZ":
        g, w = int(r.get(f"Grid_Size_{ax}", 1) or 1), int(r.get(f"Workgroup_Size_{ax}", 1) or 1)
        wg *= max(1, g // max(1, w))
    k = "<32 WGs" if wg < 32 else "32-255" if wg < 256 else "256-1023" if wg < 1024 else ">=1024"
    bins[k][0] += 1; bins[k][1] += d
    if wg >= 256:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"void ", "", name)[:60]
        byname[name][0] += 1; byname[name][1] += d
tot = sum(v[1] for v in bins.values())
for k, (n, t) in bins.items():
    print(f"{k:10s} n={n:5d}  {t/1e3:7.2f} ms  ({100*t/tot:4.1f} %)")
print("kernels with >= 256 workgroups:")
for k, (n, t) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t/1e3:7.3f} ms n={n:4d} avg {t/n:7.1f} us  {k}")
