#!/bin/bash
# marching forward with 32 input channels (NB = 4) against the lean kernel:  PASSION_MARCH_K84=0|1
for spec in "32 16 64 2" "32 8 128 1" "32 8 128 4" "32 16 128 2" "32 16 32 4"; do
  set -- $spec
  for cfg in "PASSION_MARCH_K84=0" "PASSION_MARCH_K84=1"; do
    echo "== cin $1 cout $2 S $3 B $4 $cfg"
    env MB_BATCH=$4 $cfg python3 tools/conv_microbench.py fwd $1 $2 $3 3 1 40 2>&1 | grep -E "fwd_k3|dgrad|finish|wgrad"
  done
done
