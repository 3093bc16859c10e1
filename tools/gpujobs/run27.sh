cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 900 python tools/determinism_check.py 64 passion train 40 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | tail -4; }
run PASSION_DBG_NOREC=1
run PASSION_DBG_NOREC=1 PASSION_INTRA_STREAMS=0
run PASSION_DBG_NOREC=1 PASSION_SIDE_STREAMS=1
run PASSION_DBG_X=1
run PASSION_DBG_X=2
