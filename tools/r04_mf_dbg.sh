#!/bin/bash
for dbg in 0 8 2 4 6 7; do
  echo "== marchfinal=1 dbg=$dbg: $(MB_BATCH=4 PASSION_CONV_MARCHFINAL=1 PASSION_CONV_DBG=$dbg python3 tools/conv_microbench.py bwd 8 8 128 3 1 60 2>&1 | grep -E 'dgrad')"
done
for dbg in 0 2 4 6 7; do
  echo "== marchfinal=0 dbg=$dbg: $(MB_BATCH=4 PASSION_CONV_MARCHFINAL=0 PASSION_CONV_DBG=$dbg python3 tools/conv_microbench.py bwd 8 8 128 3 1 60 2>&1 | grep -E 'dgrad')"
done
