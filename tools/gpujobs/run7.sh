cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 4 3 6 7; do
echo "=== PASSION_CONV_DBG=$dbg"
PASSION_CONV_DBG=$dbg timeout 200 python tools/conv_microbench.py fwd 8 8 128 2>&1 | tail -1
done
