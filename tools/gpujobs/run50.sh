cd $GRAFT_REPO_ROOT
timeout 900 python tools/linear_bias_determinism.py 3000 2>&1 | grep -v amdgpu | grep -v "torch sum" | grep "concurrent=True"
for i in 1 2 3; do
timeout 900 python tools/determinism_check.py 64 plain train 60 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-150 | awk '/differ/'
done
timeout 900 python tools/determinism_check.py 64 passion train 60 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-150 | awk '/differ/'
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
REPS=3 STEPS=100 bash tools/ab_bench.sh "grouped:PASSION_GROUPED_ENC=1" "streams:PASSION_GROUPED_ENC=0"
REPS=2 STEPS=60 bash tools/ab_bench.sh "passion_grouped:PASSION_GROUPED_ENC=1" "passion_streams:PASSION_GROUPED_ENC=0" -- --workload mmformer128_passion
