cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
for m in 0; do
echo "=== lean kernels (march off for level-1 shapes: A/B of the flat->global change against the earlier 43.2/76.7/72.5/77.9 us)"
PASSION_CONV_MARCH=0 timeout 200 python tools/conv_microbench.py bwd 8 8 128 2>&1 | tail -3
PASSION_CONV_MARCH=0 timeout 200 python tools/conv_microbench.py bwd 16 8 128 2>&1 | tail -3
done
echo "=== level 3 / 4 / 2(Cin 32) / 1x1x1 32->8 @128"
timeout 200 python tools/conv_microbench.py bwd 32 32 32 2>&1 | tail -3
timeout 200 python tools/conv_microbench.py bwd 64 64 16 2>&1 | tail -3
timeout 200 python tools/conv_microbench.py bwd 32 16 64 2>&1 | tail -3
timeout 200 python tools/conv_microbench.py bwd 32 8 128 1 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --workload mmformer128_passion 2>/dev/null | cut -c1-200
