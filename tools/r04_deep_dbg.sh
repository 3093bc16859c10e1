#!/bin/bash
for spec in "32 32 32 4" "128 64 16 4"; do
  set -- $spec
  for dbg in 0 4 6 7 23 31 2 8 1; do
    echo "== cin $1 cout $2 S $3 B $4 dbg=$dbg: $(MB_ONESEG=1 MB_BATCH=$4 PASSION_CONV_DBG=$dbg python3 tools/conv_microbench.py fwd $1 $2 $3 3 1 200 2>&1 | grep fwd_k3)"
  done
done
echo "== empty-ish kernel reference: 1x1 8->8 at 8^3"
MB_BATCH=1 python3 tools/conv_microbench.py fwd 8 8 8 1 1 200 2>&1 | grep fwd_k
