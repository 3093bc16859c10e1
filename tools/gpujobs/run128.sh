cd $GRAFT_REPO_ROOT
python tools/slow_launches.py 2>&1 | grep -v Warn | grep -i "c1\|timed"
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160; done
