set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_flash_attn_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/attn_microbench.py --drop 0.1 2>&1 | tail -1
timeout 300 python tools/attn_microbench.py --drop 0.0 2>&1 | tail -1
timeout 300 python tools/attn_microbench.py --drop 0.0 --no-bwd 2>&1 | tail -1
timeout 300 python tools/attn_microbench.py --drop 0.1 --no-bwd 2>&1 | tail -1
timeout 300 python tools/attn_microbench.py --B 1 --N 2048 --drop 0.1 2>&1 | tail -1
timeout 600 bash tools/pmc_attention.sh gpurun_out/r02b_attention_pmc.json --drop 0.1 > gpurun_out/pmc_attn_new.log 2>&1; echo "pmc rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02b_attention_pmc.json'))
for k,v in d['kernels'].items(): print(k[:60], v)
print(d['all_attention_kernels_mfma_busy_frac'])
PY
