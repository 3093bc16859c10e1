#!/bin/bash
# HBM-side traffic of the conv kernel families of one bench.py step, per launch, as the MI355X guide prescribes:
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes (no trace domains), FETCH_SIZE doubled on gfx950 (it tallies the
# 128-byte requests of wide coalesced reads at 64 bytes), both in KiB.   usage: tools/pmc_traffic.sh OUT.json [bench args]
OUT=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/pmc_traffic
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_traffic/$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-dice "$@" > $R/gpurun_out/pmc_traffic/$C.log 2>&1
done
cd $R
python3 - "$OUT" "$@" <<'PY'
import csv, glob, collections, json, re, sys
fam = collections.defaultdict(lambda: collections.defaultdict(list))
def family(k):                      # kernel instantiation, spelled like the names bench.py gets from the library
    m = re.search(r"((conv_march_kernel|conv_pair_kernel|conv_deep_kernel|conv_pw8_fwd_kernel|conv_pw8_dgrad_kernel|conv_pw_fwd_kernel|conv_pw_dgrad_kernel|conv_s2_dgrad_kernel|conv_lean_kernel_occ4|conv_lean_kernel|wgrad_lean_kernel|wgrad_march_kernel|conv_bf16_kernel|conv_wgrad_bf16_kernel)<[0-9a-z, ]+>)", k)
    if not m and "conv_s2_march_kernel" in k:
        return "conv_s2_march_kernel"
    return m.group(1) if m else None
for f in glob.glob("gpurun_out/pmc_traffic/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        fm = family(r["Kernel_Name"])
        if fm:
            fam[fm][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"unit": "bytes per launch, averaged over that kernel's launches in the traced training steps OF THIS WORKLOAD (bench.py reads the file of its own workload: the average is over the same set of launches as its algorithmic bytes per launch)",
       "bench_args": " ".join(sys.argv[2:]),
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE x2 (gfx950)",
       "bytes_per_launch": {}, "fetch_bytes": {}, "write_bytes": {}, "launches": {}}
for fm, d in fam.items():
    fe = 2 * 1024 * sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
    wr = 1024 * sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    out["fetch_bytes"][fm], out["write_bytes"][fm] = round(fe), round(wr)
    out["bytes_per_launch"][fm] = round(fe + wr)
    out["launches"][fm] = len(d["FETCH_SIZE"])
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/pmc_traffic/FETCH_SIZE gpurun_out/pmc_traffic/WRITE_SIZE
