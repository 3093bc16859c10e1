#!/bin/bash
# fused attention fwd+bwd on the four shapes of the benchmarked steps (plain: B=1 N=2048, B=4 N=512; PASSION: B=5 N=2048, B=20 N=512)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "1 2048" "4 512" "5 2048" "20 512"; do set -- $shape; python tools/attn_microbench.py --B $1 --N $2 --iters 300 2>&1 | tail -1; done
