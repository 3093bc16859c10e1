cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "march" 2>&1 | tail -3
for tw in 0 16 32; do
echo "=== PASSION_MARCH_TW=$tw"
PASSION_MARCH_TW=$tw timeout 200 python tools/conv_microbench.py bwd 8 8 128 2>&1 | tail -3
PASSION_MARCH_TW=$tw timeout 200 python tools/conv_microbench.py bwd 16 8 128 2>&1 | tail -3
done
for dbg in 7; do
echo "=== PASSION_CONV_DBG=$dbg"
PASSION_CONV_DBG=$dbg timeout 200 python tools/conv_microbench.py fwd 8 8 128 2>&1 | tail -1
done
PASSION_MARCH_PLANES=8 timeout 200 python tools/conv_microbench.py bwd 16 16 64 2>&1 | tail -3
PASSION_MARCH_PLANES=8 PASSION_MARCH_TW=16 timeout 200 python tools/conv_microbench.py bwd 16 16 64 2>&1 | tail -3
