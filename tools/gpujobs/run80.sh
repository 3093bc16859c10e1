cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | grep -E "^E  |passed|failed" | head -6
done
PASSION_FUSE_STREAM=0 timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | grep -E "^E  |passed|failed" | head -6
