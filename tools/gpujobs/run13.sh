cd $GRAFT_REPO_ROOT
timeout 600 python tools/depth_debug.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider\|print(" | cut -c1-400
timeout 1200 python -m pytest tests/test_depth_gpu.py -x -q -s -m gpu 2>&1 | tail -15 | cut -c1-250
