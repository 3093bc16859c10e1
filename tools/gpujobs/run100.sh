cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgrad_march" 2>&1 | tail -3
for nb in 1 4; do
for dbg in 0 1 2 32 ; do
echo "== batch $nb dbg $dbg"
MB_BATCH=$nb PASSION_CONV_DBG=$dbg python tools/conv_microbench.py bwd 8 8 128 3 1 30 2>&1 | grep wgrad
done
done
MB_BATCH=4 python tools/conv_microbench.py bwd 16 8 128 3 1 30 2>&1 | grep wgrad
MB_BATCH=4 PASSION_WGRAD_MARCH=0 python tools/conv_microbench.py bwd 16 8 128 3 1 30 2>&1 | grep wgrad
