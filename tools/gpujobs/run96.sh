cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | tail -1 | cut -c1-120
for m in 1 0 1 0; do
PASSION_WGRAD_MARCH=$m timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r96_$m.json
python - <<PY
import json
j=json.load(open("gpurun_out/r96_$m.json")); r=j["roofline"]
print($m, j["value"], j["ms_per_step"], r["kernel"][:28], r["avg_launch_ms_single_stream"], r["frac_single_stream"], r["avg_launch_ms"], r["frac"], r["ms_per_step_by_family"])
PY
done
