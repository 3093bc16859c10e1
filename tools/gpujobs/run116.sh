cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "first_layer or raw_input or grouped" 2>&1 | tail -15
python tools/slow_launches.py 2>&1 | grep -v Warn | grep -i "c1\|timed"
