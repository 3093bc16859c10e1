cd $GRAFT_REPO_ROOT
timeout 900 python tools/determinism_check.py 64 passion train 8 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | tail -30
echo == plain
timeout 900 python tools/determinism_check.py 64 plain train 8 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | tail -8
