cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 1500 python tools/ddp_debug.py f32 12 2>&1 | grep "^trial" | awk '{print $3}' | tr '\n' ' '; echo; }
run DDP_OVERLAP=0
run DDP_BUCKET_MB=32
run DDP_BUCKET_MB=0.05
