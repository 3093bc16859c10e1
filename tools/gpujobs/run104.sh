cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 3 4 6; do
echo "== fwd batch 4 dbg $dbg"
MB_BATCH=4 PASSION_CONV_DBG=$dbg python tools/conv_microbench.py fwd 8 8 128 3 1 30 2>&1 | grep "fwd_k3"
done
for dbg in 0 1 2 4; do
echo "== bwd batch 4 dbg $dbg"
MB_BATCH=4 PASSION_CONV_DBG=$dbg python tools/conv_microbench.py bwd 8 8 128 3 1 30 2>&1 | grep "dgrad"
done
