#!/bin/bash
# rocprofv3 PMC passes (separate passes, no trace domains combined) over one micro-benchmark command
# usage: tools/pmc_run.sh OUTDIR -- python3 tools/conv_microbench.py ...
OUT=$1; shift; shift
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=()
for a in "$@"; do case "$a" in tools/*) ARGS+=("$R/$a");; *) ARGS+=("$a");; esac; done
set -- "${ARGS[@]}"
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d $R/$OUT/pass$i -- "$@" > $R/$OUT/pass$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv" not in k and "wgrad" not in k: continue
        agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
