cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160; done
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --workload mmformer128_passion 2>/dev/null | tail -1 | cut -c1-160; done
timeout 900 python tools/determinism_check.py 128 passion train 30 2>&1 | grep -v Warn | tail -2
