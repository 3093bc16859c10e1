#!/bin/bash
# pipelined step time under a few HIP runtime settings (size 128): which ones move the launch-latency floor?
run() { echo "== $*"; env "$@" python tools/sync_check.py ${SIZE:-128} 2>&1 | grep "^pipelined:" ; }
run A=1
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_SKIP_RELEASE_SCOPE=1
run AMD_OPT_FLUSH=0
run DEBUG_HIP_DYNAMIC_QUEUES=0
run ROC_SYSTEM_SCOPE_SIGNAL=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run PASSION_ENC_STREAMS=0
run A=2
