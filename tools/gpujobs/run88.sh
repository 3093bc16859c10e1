cd $GRAFT_REPO_ROOT
timeout 600 python tools/null_stream_event.py 2>&1 | grep -v amdgpu
