cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/prof_tl
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 3 > $R/gpurun_out/prof_tl/bench.json 2> $R/gpurun_out/prof_tl/bench.err)
f=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f | head -12
python - "$f" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id","")) for r in rows)
idx = [i for i,e in enumerate(ev) if "adamw_multi_kernel" in e[2]]
groups=[]
for i in idx:
    if not groups or i-groups[-1][-1]>50: groups.append([i])
    else: groups[-1].append(i)
k=len(groups)-3
step=ev[groups[k-1][-1]+1:groups[k][-1]+1]
by=collections.defaultdict(lambda: collections.defaultdict(float))
for s,e,n,st in step:
    n=re.sub(r"\(anonymous namespace\)::|void |at::native::|_GLOBAL__N_1","",n)[:60]
    by[st][n]+=e-s
for st,d in sorted(by.items(), key=lambda x:-sum(x[1].values())):
    print(f"stream {st}: {sum(d.values())/1e6:.2f} ms, {len(d)} kernel kinds")
    for n,v in sorted(d.items(), key=lambda x:-x[1])[:14]:
        print(f"     {v/1e6:6.3f} ms  {n}")
PY
rm -rf gpurun_out/prof_tl
