cd $GRAFT_REPO_ROOT
PASSION_DDP_DEBUG=1 timeout 1500 python tools/ddp_debug.py f32 6 > gpurun_out/ddp_debug.log 2>&1
grep -n "^trial\|  trial" gpurun_out/ddp_debug.log | head -20
