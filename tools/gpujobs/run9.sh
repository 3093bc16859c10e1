cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "march or conv_block or kernel_paths" 2>&1 | tail -3
for tw in 0 32; do
echo "=== PASSION_MARCH_TW=$tw"
PASSION_MARCH_TW=$tw timeout 200 python tools/conv_microbench.py bwd 8 8 128 2>&1 | tail -3
PASSION_MARCH_TW=$tw timeout 200 python tools/conv_microbench.py bwd 16 8 128 2>&1 | tail -3
done
for dbg in 1 2 4 7; do
echo "=== PASSION_CONV_DBG=$dbg"
PASSION_CONV_DBG=$dbg timeout 200 python tools/conv_microbench.py bwd 8 8 128 2>&1 | tail -2 | head -1
done
