cd $GRAFT_REPO_ROOT
python tools/slow_launches.py passion 2>&1 | grep -v Warn | head -32
