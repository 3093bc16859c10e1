cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "statistics or colsum or linear_fn" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_depth_gpu.py tests/test_benchsize_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/bench_r02f_plain.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r02f_plain.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['ms_per_step_by_family'])
PY
PASSION_FUSE_STATS=0 timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200
