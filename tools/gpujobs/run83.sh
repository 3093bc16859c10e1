cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | grep -E "AssertionError: \(|passed|failed" | cut -c1-330 | head -2
done
