cd $GRAFT_REPO_ROOT
timeout 1200 python tools/ddp_debug.py f32 5 2>&1 | grep -v "amdgpu\|Warning" | tail -40
