cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8
cat gpurun_out/bench_r02d_passion.json 2>/dev/null | tail -c 1500
timeout 600 python bench.py --no-cpu-baseline --passion 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/bench_r02d_passion.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r02d_passion.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['ms_per_step_by_family'])
PY
