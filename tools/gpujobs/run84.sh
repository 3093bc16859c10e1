cd $GRAFT_REPO_ROOT
timeout 1500 python tools/ddp_debug.py f32 14 2>&1 | grep "trial" | tail -60
