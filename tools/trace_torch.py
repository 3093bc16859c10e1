"""torch (at::native) kernels of the LAST training step with their full names, grids and times"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "adamw_multi_kernel" in r["Kernel_Name"]]
ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] - i > 200]
if len(ends) >= 2:
    rows = rows[ends[-2] + 1: ends[-1] + 1]
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for r in rows:
    n = r["Kernel_Name"]
    if "at::native" not in n and "rocclr" not in n:
        continue
    m = re.search(r"at::native::(\w+)<[^>]*?(at::native::)?(\w+Functor\w*|\w+_kernel_cuda|\w+KernelImpl|\w+kernel)\b", n)
    key = re.sub(r"std::array<char\*, \d+ul>|unsigned int|\(anonymous namespace\)::", "", n)[:150]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1; agg[key][1] += d; agg[key][2][r.get("Grid_Size_X", "?")] += 1
for k, (c, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e3:7.3f} ms n={c:4d} avg {t/c:7.1f} us grids {dict(g.most_common(3))}\n      {k}")
