cd $GRAFT_REPO_ROOT
run() { echo "== $*"; for i in 1 2 3 4; do env "$@" timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | grep -E "AssertionError: \(|passed|failed" | cut -c1-120 | head -2; done; }
run PASSION_GROUPED_INTRA=0
run PASSION_GROUPED_ENC=0
run PASSION_FUSE_STATS=0
run PASSION_GEMM_1X1=0
