cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "first_layer or raw_input or grouped" 2>&1 | tail -2
python tools/slow_launches.py 2>&1 | grep -v Warn | grep -i "c1\|timed"
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160; done
