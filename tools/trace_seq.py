"""every dispatch of the LAST training step of a rocprofv3 --kernel-trace CSV in start order: offset from the step's first kernel (us),
duration (us), queue, workgroups, kernels already in flight when it starts, name.  python tools/trace_seq.py trace.csv"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
c1 = [i for i, r in enumerate(rows) if ("conv_c1_fwd_kernel" in r["Kernel_Name"] or "conv_c1m_fwd_kernel" in r["Kernel_Name"])]
opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "adamw_multi_kernel" in r["Kernel_Name"]]
ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] - i > 200]
if len(c1) >= 3:          # a step = from one first-layer forward to the next (the mmformer workloads)
    rows = rows[c1[-2]: c1[-1]]
elif len(ends) >= 2:
    rows = rows[ends[-2] + 1: ends[-1] + 1]
t0 = int(rows[0]["Start_Timestamp"])
qs = {}
live = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    live = [x for x in live if x > s]
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"void |pconv::|at::native::", "", name)
    name = re.sub(r"\(.*", "", name)[:70]
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    wg = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) * int(r.get("Grid_Size_Y", "1") or 1) * int(r.get("Grid_Size_Z", "1") or 1) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1) * int(r.get("Workgroup_Size_Y", "1") or 1) * int(r.get("Workgroup_Size_Z", "1") or 1))
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} wg{wg:6d} +{len(live)} {name}")
    live.append(e)
