cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "bit_for_bit" 2>&1 | grep -E "^E  |passed|failed" | head -8
done
