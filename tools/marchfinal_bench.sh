#!/bin/bash
# finest-level 3x3x3 backward with the finish-free path on / off (PASSION_CONV_MARCHFINAL), one layer at a time
for mf in 1 0; do
  for spec in "8 8 128 4" "16 8 128 4" "8 8 128 1"; do
    set -- $spec
    echo "== cin $1 cout $2 S $3 B $4 marchfinal=$mf"
    MB_BATCH=$4 PASSION_CONV_MARCHFINAL=$mf python3 tools/conv_microbench.py bwd $1 $2 $3 3 1 60 2>&1 | grep -v "amdgpu.ids"
  done
done
