cd $GRAFT_REPO_ROOT
for mode in "passion train 12 blobs" "passion train 12 synthetic" "passion eval 12 blobs" "plain train 12 blobs"; do
echo "== $mode"
timeout 600 python tools/determinism_check.py 64 $mode 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | head -30
done
echo "== no side streams"
PASSION_ENC_STREAMS=0 timeout 600 python tools/determinism_check.py 64 passion train 12 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | head -30
