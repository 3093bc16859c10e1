cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -1; done
for i in 1 2 3; do timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "repeats_bit_for_bit" 2>&1 | tail -1; done
