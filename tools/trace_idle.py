"""GPU busy/idle of the LAST training step of a rocprofv3 --kernel-trace CSV (union of kernel intervals over all streams)
and launches by count: python tools/trace_idle.py trace.csv [top]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "adamw_multi_kernel" in r["Kernel_Name"]]
ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] - i > 200]
if len(ends) >= 2:
    rows = rows[ends[-2] + 1: ends[-1] + 1]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
ksum = sum(e - s for s, e in iv)
print(f"{len(rows)} dispatches, span {span/1e6:.2f} ms, busy (union) {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms in {len(gaps)} gaps "
      f"(median {sorted(gaps)[len(gaps)//2]/1e3:.1f} us), sum of kernel durations {ksum/1e6:.2f} ms")
q = collections.Counter(r.get("Queue_Id", "?") for r in rows)
print("queues:", dict(q))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"void ", "", name)[:90]
    agg[name][0] += 1; agg[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"n={c:5d} {t/1e3:8.3f} ms avg {t/c:7.1f} us  {k}")
