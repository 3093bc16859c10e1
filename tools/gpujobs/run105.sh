cd $GRAFT_REPO_ROOT
python tools/host_profile.py plain 20 2>&1 | grep -v Warn | head -60
