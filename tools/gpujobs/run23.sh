cd $GRAFT_REPO_ROOT
timeout 600 python tools/determinism_check.py 64 passion train 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | head -60
