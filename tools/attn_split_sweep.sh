#!/bin/bash
# split-K sweep of the fused attention on the shapes of a plain step (B=1 N=2048 inter-modal, B=4 N=512 intra-modal) and of a
# PASSION step (B=5 N=2048, B=20 N=512)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "1 2048" "4 512" "5 2048" "20 512" "2 2048"; do
  set -- $shape
  for sp in 1 2 4 8; do
    echo -n "split=$sp "; PASSION_FLASH_SPLIT=$sp python tools/attn_microbench.py --B $1 --N $2 --iters 200 2>&1 | tail -1
  done
done
