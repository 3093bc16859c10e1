cd $GRAFT_REPO_ROOT
timeout 600 python tools/encoder_group_probe.py 128 2>&1 | grep -v amdgpu
timeout 600 python tools/encoder_group_probe.py 128 2>&1 | grep -v amdgpu
