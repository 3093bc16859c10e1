cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
PASSION_DEVICE_INDEX=0 PASSION_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
