cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 900 python tools/determinism_check.py 64 plain train 24 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-150 | awk 'NR<=3 || /pairs differ/'; }
run PASSION_X=1
run PASSION_ENC_STREAMS=0
run PASSION_INTRA_STREAMS=0
run PASSION_GROUPED_ENC=0
