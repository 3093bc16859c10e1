#!/bin/bash
# MFMA utilisation of the attention kernels (north_star: "rocprof MFMA-utilisation reported"): rocprofv3 counter passes (no trace
# domains combined with --pmc) over tools/attn_microbench.py, plus one --kernel-trace pass for the durations.
#   usage: tools/pmc_attention.sh OUT.json [attn_microbench args]
# busy fraction of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8
# (rocprofv3 reports the sum over the 8 XCDs, MI355X guide 'DVFS give-back'); SQ_VALU_MFMA_BUSY_CYCLES counts cycles
# (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32).
OUT=$1; shift
R=$PWD
D=$R/gpurun_out/pmc_attn
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $D/pmc -- python3 $R/tools/attn_microbench.py --iters 3 "$@" > $D/pmc.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $D/pmc2 -- python3 $R/tools/attn_microbench.py --iters 3 "$@" > $D/pmc2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/tools/attn_microbench.py --iters 10 "$@" > $D/trace.log 2>&1
cd $R
python3 - "$OUT" "$D" "$*" <<'PY'
import csv, glob, collections, json, sys
out_path, D, args = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(D + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("attn", "bgemm", "softmax", "dropout", "flash")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(D + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
res = {"method": "rocprofv3 --pmc passes over tools/attn_microbench.py " + args + "; busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
       "kernels": {}}
tot_busy = tot_cyc = 0.0
for k, d in agg.items():
    m = lambda c: (sum(d[c]) / len(d[c])) if d.get(c) else None
    cyc = (m("GRBM_GUI_ACTIVE") or 0) / 8.0
    busy = m("SQ_VALU_MFMA_BUSY_CYCLES") or 0.0
    e = {"launches_profiled": len(d.get("GRBM_GUI_ACTIVE", [])), "kernel_cycles": round(cyc), "mfma_busy_cycles_all_simds": round(busy),
         "mfma_busy_frac": round(busy / (1024.0 * cyc), 4) if cyc else None}
    for c in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_VALU_MFMA_COEXEC_CYCLES",
              "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if m(c) is not None:
            e[c] = round(m(c))
    for name, (calls, avg, tot) in dur.items():
        if name == k:
            e["avg_us"] = round(avg / 1e3, 2)
    n = len(d.get("GRBM_GUI_ACTIVE", [])) or 1
    tot_busy += busy * n
    tot_cyc += cyc * n
    res["kernels"][k[:120]] = e
res["all_attention_kernels_mfma_busy_frac"] = round(tot_busy / (1024.0 * tot_cyc), 4) if tot_cyc else None
import hashlib, os
res["flash_attn_sha256"] = hashlib.sha256(open(os.path.join("passion_amd", "csrc", "flash_attn.hip"), "rb").read()).hexdigest()   # bench.py refuses the file for any other source
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
