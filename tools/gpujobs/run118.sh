cd $GRAFT_REPO_ROOT
timeout 900 python tools/determinism_check.py 128 plain train 60 2>&1 | grep -v Warn | tail -4
timeout 900 python tools/determinism_check.py 128 passion train 40 2>&1 | grep -v Warn | tail -4
timeout 900 python tools/determinism_check.py 64 passion train 80 2>&1 | grep -v Warn | tail -3
