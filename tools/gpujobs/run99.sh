cd $GRAFT_REPO_ROOT
for nb in 1 4; do
for dbg in 0 1 2 8 16 32 33 ; do
echo "== batch $nb dbg $dbg"
MB_BATCH=$nb PASSION_CONV_DBG=$dbg python tools/conv_microbench.py bwd 8 8 128 3 1 30 2>&1 | grep wgrad
done
echo "== batch $nb lean"
MB_BATCH=$nb PASSION_WGRAD_MARCH=0 python tools/conv_microbench.py bwd 8 8 128 3 1 30 2>&1 | grep wgrad
done
