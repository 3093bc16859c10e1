cd $GRAFT_REPO_ROOT
timeout 600 python tools/pw_bench.py 2>&1 | grep -v amdgpu
