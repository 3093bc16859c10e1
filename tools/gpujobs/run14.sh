cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_depth_gpu.py -x -q -s -m gpu 2>&1 | grep -v "amdgpu\|Gloo\|Warn\|Consider" | grep "passed\|failed\|depth\|distance\|Error\|assert" | head -20 | cut -c1-250
