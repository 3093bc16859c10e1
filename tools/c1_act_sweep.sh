#!/bin/bash
# first-layer weight gradient (wgrad_c1_kernel): its tests, then its average duration in the benchmark step for a few counts of
# summing workgroups (PASSION_C1_ACT)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "first_layer" 2>&1 | tail -3
for a in ${ACTS:-256 320}; do
  PASSION_C1_ACT=$a bash tools/kernel_ab.sh c1act$a >/dev/null 2>&1
  echo "ACT=$a: $(grep -i "wgrad_c1" gpurun_out/kab_c1act$a.txt | head -2)"
done
