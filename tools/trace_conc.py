"""concurrency profile of the LAST training step of a rocprofv3 --kernel-trace CSV: time with 0 / 1 / 2 / 3+ kernels in flight, and the
kernels that run ALONE (nothing else in flight) ranked by their lonely time -- the critical path candidates.
python tools/trace_conc.py trace.csv [top]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
opt = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r["Kernel_Name"]]
ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] - i > 200]
if len(ends) >= 2:
    rows = rows[ends[-2] + 1: ends[-1] + 1]
ev = []
for k, r in enumerate(rows):
    ev.append((int(r["Start_Timestamp"]), 1, k)); ev.append((int(r["End_Timestamp"]), -1, k))
ev.sort()
live = set()
hist = collections.Counter()
alone = collections.defaultdict(float)
last = ev[0][0]
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)[:70]
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        hist[min(len(live), 3)] += dt
        if len(live) == 1:
            alone[short(rows[next(iter(live))]["Kernel_Name"])] += dt
    last = t
    if d == 1: live.add(k)
    else: live.discard(k)
span = ev[-1][0] - ev[0][0]
print(f"{len(rows)} dispatches, span {span / 1e6:.3f} ms")
for c in sorted(hist):
    print(f"  {c}{'+' if c == 3 else ' '} kernels in flight: {hist[c] / 1e6:7.3f} ms ({100 * hist[c] / span:5.1f} %)")
print("kernels by the time they run with nothing beside them:")
for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:top]:
    print(f"  {v / 1e3:8.1f} us  {n}")
