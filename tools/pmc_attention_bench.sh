#!/bin/bash
# MFMA busy fraction of the attention kernels AS THE BENCHMARK RUNS THEM (the microbench of tools/pmc_attention.sh is
# another instantiation): rocprofv3 counter pass over bench.py itself (program directly after `--`), flash_* kernels only.
#   usage: tools/pmc_attention_bench.sh OUT.json [bench args, e.g. --workload mmformer128_passion]
OUT=$1; shift
R=$PWD
D=$R/gpurun_out/pmc_attn_bench
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $D/pmc -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-dice --steps 5 --warmup 2 --prime 2 "$@" > $D/pmc.log 2>&1
cd $R
python3 - "$OUT" "$D" "$*" <<'PY'
import csv, glob, collections, json, sys
out_path, D, args = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(D + "/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "flash_q_kernel" in k or "flash_k_kernel" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ... -- python3 bench.py --no-cpu-baseline --steps 5 " + args +
                 "; busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), every flash_* launch of the run", "kernels": {}}
tb = tc = 0.0
for k, d in sorted(agg.items()):
    cyc = [v / 8.0 for v in d["GRBM_GUI_ACTIVE"]]
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"]
    n = min(len(cyc), len(busy))
    res["kernels"][k.replace("(anonymous namespace)::", "")[:100]] = {
        "launches": n, "avg_kernel_cycles": round(sum(cyc[:n]) / n), "mfma_busy_frac": round(sum(busy[:n]) / (1024.0 * sum(cyc[:n])), 4),
        "valu_insts_per_launch": round(sum(d["SQ_INSTS_VALU"]) / max(1, len(d["SQ_INSTS_VALU"])))}
    tb += sum(busy[:n]); tc += sum(cyc[:n])
res["all_flash_kernels_mfma_busy_frac"] = round(tb / (1024.0 * tc), 4) if tc else None
import hashlib, os
res["flash_attn_sha256"] = hashlib.sha256(open(os.path.join("passion_amd", "csrc", "flash_attn.hip"), "rb").read()).hexdigest()   # bench.py refuses the file for any other source
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
