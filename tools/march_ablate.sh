#!/bin/bash
# where a marching launch spends its time: PASSION_CONV_DBG bits 1 (no staging) / 2 (no tap loop) / 4 (no epilogue)
for mode in bwd fwd; do
  for dbg in 0 1 2 4 3 6 5 7; do
    echo "== $mode 8 -> 8, 4 x 128^3, dbg $dbg"
    MB_BATCH=4 PASSION_CONV_DBG=$dbg python3 tools/conv_microbench.py $mode 8 8 128 3 1 40 2>&1 | grep -E "fwd_k3|dgrad|finish|wgrad"
  done
done
