set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_flash_attn_gpu.py -x -q -s -m gpu > gpurun_out/r02_flash.log 2>&1; echo "flash rc $?"
tail -40 gpurun_out/r02_flash.log
timeout 300 python tools/attn_microbench.py --drop 0.1 2>&1 | tail -2
timeout 300 python tools/attn_microbench.py --drop 0.0 2>&1 | tail -2
timeout 300 python tools/attn_microbench.py --drop 0.0 --no-bwd 2>&1 | tail -2
timeout 300 python tools/attn_microbench.py --drop 0.1 --no-bwd 2>&1 | tail -2
timeout 300 python tools/attn_microbench.py --B 1 --N 512 --drop 0.1 2>&1 | tail -2
