cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 900 python tools/determinism_check.py 64 plain train 60 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-170 | awk 'NR<=6 || /pairs differ/'
done
for i in 1 2; do
timeout 900 python tools/determinism_check.py 64 passion train 60 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-170 | awk 'NR<=6 || /pairs differ/'
done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
