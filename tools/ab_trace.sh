#!/bin/bash
# single-stream kernel totals of the last step for two builds of the library: tools/ab_trace.sh "pattern|pattern"
cd /tmp && export TMPDIR=/tmp
for l in libpassion_hip_prev.so libpassion_hip.so; do
  export PASSION_HIP_LIB=$GRAFT_REPO_ROOT/passion_amd/lib/$l
  PASSION_ENC_STREAMS=0 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/trab -o tr --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --prime 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/trab.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/trab -name "*kernel_trace.csv" | head -1)
  echo $l; python3 $GRAFT_REPO_ROOT/tools/trace_last.py $f 80 | grep -E "last step|$1"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/trab
done
