cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 1500 python tools/ddp_debug.py f32 12 2>&1 | grep "^trial" | awk '{print $3}' | tr '\n' ' '; echo; }
run PASSION_GROUPED_INTRA=0
run PASSION_GEMM_1X1=0
run PASSION_FUSE_STATS=0
run PASSION_ENC_STREAMS=0
