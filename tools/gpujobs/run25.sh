cd $GRAFT_REPO_ROOT
timeout 600 python tools/linear_bias_determinism.py 2>&1 | grep -v amdgpu | tail
