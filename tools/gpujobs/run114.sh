cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "upsample" 2>&1 | tail -3
python tools/upsample_bench.py 2>&1 | grep -v Warn
