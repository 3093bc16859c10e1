#!/bin/bash
# deep-level conv kernels alone: conv_deep on / off
for spec in "32 32 32 4" "64 32 32 4" "64 64 16 4" "128 64 16 4" "128 128 8 4" "128 64 16 1" "64 32 32 1" "128 128 8 1"; do
  set -- $spec
  for deep in 1 0; do
    echo "== cin $1 cout $2 S $3 B $4 deep=$deep"
    MB_ONESEG=1 MB_BATCH=$4 PASSION_CONV_DEEP=$deep python3 tools/conv_microbench.py bwd $1 $2 $3 3 1 100 2>&1 | grep -v "amdgpu.ids\|^bwd"
  done
done
for deep in 1 0; do
  echo "== 1x1 cin 64 cout 64 S 16 deep=$deep"
  MB_ONESEG=1 MB_BATCH=4 PASSION_CONV_DEEP=$deep python3 tools/conv_microbench.py bwd 64 64 16 1 1 100 2>&1 | grep -v "amdgpu.ids\|^bwd"
done
