#!/bin/bash
# prefetch depth of the B fragments in the marching tap loop (A/B builds from tools/build_variant.sh pdN conv_march.hip -DMARCH_PD=N)
for spec in "8 8 128 4" "16 8 128 4" "16 16 64 4" "8 8 128 1"; do
  set -- $spec
  for pd in ${PDS:-1 2 3 4}; do
    echo "== cin $1 cout $2 S $3 B $4 PD $pd"
    MB_BATCH=$4 PASSION_HIP_LIB=$PWD/passion_amd/lib/libpassion_hip_pd$pd.so python3 tools/conv_microbench.py bwd $1 $2 $3 3 1 40 2>&1 | grep -E "fwd_k3|dgrad"
  done
done
