cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
