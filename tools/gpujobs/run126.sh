cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160
