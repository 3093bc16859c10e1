"""GPU timeline of one training step from a rocprofv3 kernel trace (csv): how much of the step has 0 / 1 / 2+ kernels in
flight, per-queue busy time, and the longest stretches with a single small kernel.  usage: timeline.py TRACE.csv [step_index]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "")) for r in rows]
ev.sort()
# steps are delimited by the optimizer kernel (adamw_multi_kernel): take the span between the last adamw of step k-1 and of step k
idx = [i for i, e in enumerate(ev) if "adamw_multi_kernel" in e[2]]
groups = []
for i in idx:
    if not groups or i - groups[-1][-1] > 50:
        groups.append([i])
    else:
        groups[-1].append(i)
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(groups) - 3
lo, hi = groups[k - 1][-1] + 1, groups[k][-1] + 1
step = ev[lo:hi]
t0, t1 = step[0][0], max(e[1] for e in step)
print(f"step {k}: {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, kernel time {sum(e[1] - e[0] for e in step) / 1e6:.3f} ms")
pts = []
for s, e, n, q, st in step:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
hist = collections.Counter()
cur, last = 0, t0
for t, d in pts:
    hist[min(cur, 4)] += t - last
    cur += d; last = t
for c in sorted(hist):
    print(f"  {c}{'+' if c == 4 else ' '} kernels in flight: {hist[c] / 1e6:7.3f} ms ({100 * hist[c] / (t1 - t0):5.1f} %)")
byq = collections.defaultdict(float)
for s, e, n, q, st in step:
    byq[(q, st)] += e - s
for q, v in sorted(byq.items(), key=lambda x: -x[1]):
    print(f"  queue/stream {q}: busy {v / 1e6:.3f} ms")
# idle gaps: where, and which kernel came before/after
gaps = []
cur, last = 0, t0
active_end = t0
for s, e, n, q, st in sorted(step):
    if s > active_end:
        gaps.append((s - active_end, active_end - t0, n))
    active_end = max(active_end, e)
gaps.sort(reverse=True)
print(f"  idle gaps: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.3f} ms; largest:")
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)[:70]
for g, at, n in gaps[:12]:
    print(f"    {g / 1e3:7.1f} us at +{at / 1e6:6.3f} ms before {short(n)}")
# phases: cumulative kernel time by 1 ms bins
bins = collections.defaultdict(float)
for s, e, n, q, st in step:
    bins[int((s - t0) / 1e6)] += e - s
print("  kernel time per 1 ms bin of the step:", " ".join(f"{bins[i] / 1e6:.1f}" for i in range(int((t1 - t0) / 1e6) + 1)))
