#!/bin/bash
# low-level conv shapes (levels 3..5) alone: forward / dgrad / wgrad durations for batch 1 and 4
cd $GRAFT_REPO_ROOT
for mt in ${MTS:-512 4096}; do
for cfg in "32 32 32" "64 64 16" "128 128 8" "256 128 8"; do
  set -- $cfg
  for nb in ${NBS:-1 4}; do
    echo "== MT_TILES=$mt cin=$1 cout=$2 S=$3 batch=$nb"
    PASSION_LEAN_MT_TILES=$mt MB_BATCH=$nb python tools/conv_microbench.py bwd $1 $2 $3 3 1 30 2>&1 | grep -v amdgpu.ids | tail -3
  done
done
done
