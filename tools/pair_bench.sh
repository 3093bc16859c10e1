#!/bin/bash
# depth-pair kernel (conv_pair.hip) against the marching kernel it replaces, one layer at a time
for spec in "8 8 128 4" "8 8 128 1" "8 8 64 4"; do
  set -- $spec
  for cfg in "PASSION_CONV_PAIR=0" "PASSION_CONV_PAIR=1" "PASSION_CONV_PAIR=1 PASSION_PAIR_TW=16" "PASSION_CONV_PAIR=1 PASSION_PAIR_TW=32" $EXTRA; do
    echo "== cin $1 cout $2 S $3 B $4 $cfg"
    env MB_BATCH=$4 $cfg python3 tools/conv_microbench.py bwd $1 $2 $3 3 1 40 2>&1 | grep -E "fwd_k3|dgrad|finish"
  done
done
