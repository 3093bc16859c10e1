cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 900 python tools/determinism_check.py 64 plain train 60 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-150 | awk 'NR<=4 || /pairs differ/'
done
timeout 900 python tools/determinism_check.py 64 passion train 40 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-150 | awk 'NR<=4 || /pairs differ/'
