cd $GRAFT_REPO_ROOT
python tools/upsample_bench.py 2>&1 | grep -v Warn
