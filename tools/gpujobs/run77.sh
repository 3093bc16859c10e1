cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_tmp.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_tmp.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'][:40], r['frac'], r['frac_single_stream'], r['launches_per_step'], r.get('with_finish'), list(d.get('attention',{}).keys())[:4])
PY
timeout 900 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-150
done
