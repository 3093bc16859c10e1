#!/bin/bash
# SQ_INSTS_VALU / SQ_INSTS_SALU / LDS / busy cycles of the conv kernels for one micro-benchmark command
# usage: tools/pmc_valu.sh OUTDIR -- python3 tools/conv_microbench.py ...
OUT=$1; shift; shift
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=()
for a in "$@"; do case "$a" in tools/*) ARGS+=("$R/$a");; *) ARGS+=("$a");; esac; done
set -- "${ARGS[@]}"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p -- "$@" > $R/$OUT/p.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv" not in k and "wgrad" not in k: continue
        agg[k[:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print(k, " waves", w)
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print(f"   {c:22s} mean={m:14.1f}  per wave {m / w:9.1f}")
PY
rm -rf $OUT/p
