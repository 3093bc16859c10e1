cd $GRAFT_REPO_ROOT
python tools/slow_launches.py 2>&1 | grep -v Warn | head -45
