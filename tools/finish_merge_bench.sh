#!/bin/bash
# one finish launch for all segments of a concatenated input against one per segment (PASSION_FINISH_MERGE=0)
for spec in "16 8 128 4 3" "16 8 128 1 3" "32 16 64 4 3" "32 16 64 1 3"; do
  set -- $spec
  for m in 1 0; do
    echo "== cin $1 cout $2 S $3 B $4 merge=$m"
    MB_BATCH=$4 PASSION_FINISH_MERGE=$m python3 tools/conv_microbench.py bwd $1 $2 $3 $5 1 40 2>&1 | grep -E "dgrad|finish"
  done
done
