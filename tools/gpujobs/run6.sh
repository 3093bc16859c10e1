set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "march" 2>&1 | tail -3
for pl in 8 16 32; do
echo "=== PASSION_MARCH_PLANES=$pl"
PASSION_MARCH_PLANES=$pl timeout 200 python tools/conv_microbench.py bwd 8 8 128 2>&1 | tail -3
PASSION_MARCH_PLANES=$pl timeout 200 python tools/conv_microbench.py bwd 16 8 128 2>&1 | tail -3
PASSION_MARCH_PLANES=$pl timeout 200 python tools/conv_microbench.py bwd 16 16 64 2>&1 | tail -3
done
