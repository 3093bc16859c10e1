cd $GRAFT_REPO_ROOT
timeout 1500 python tools/ddp_debug.py f32 16 2>&1 | grep "^trial" | awk '{print $3}' | tr '\n' ' '; echo
timeout 1500 python tools/ddp_debug.py bf16 8 2>&1 | grep "^trial" | awk '{print $3}' | tr '\n' ' '; echo
for i in 1 2 3; do timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | head -2; done
