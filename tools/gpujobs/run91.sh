cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 900 python tools/determinism_check.py 64 plain train 60 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-170 | awk 'NR<=4 || /pairs differ/'
timeout 900 python tools/determinism_check.py 64 passion train 60 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-170 | awk 'NR<=4 || /pairs differ/'
done
timeout 1500 python tools/ddp_debug.py bf16 10 2>&1 | grep "^trial" | awk '{print $3}' | tr '\n' ' '; echo
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -3
