set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "march" > gpurun_out/r02_march.log 2>&1; echo "march rc $?"
tail -25 gpurun_out/r02_march.log
for m in 1 0; do
echo "=== PASSION_CONV_MARCH=$m"
PASSION_CONV_MARCH=$m timeout 200 python tools/conv_microbench.py bwd 8 8 128 2>&1 | tail -4
PASSION_CONV_MARCH=$m timeout 200 python tools/conv_microbench.py bwd 16 8 128 2>&1 | tail -4
PASSION_CONV_MARCH=$m timeout 200 python tools/conv_microbench.py bwd 16 16 64 2>&1 | tail -4
done
