#!/bin/bash
# bench lines of every workload, graph and eager launch
TAG=${1:-r04}
R=$PWD
for wl in mmformer128 mmformer128_passion rfnet128_passion m2ftrans128 m2ftrans128_passion; do
  for mode in graph eager; do
    BENCH_HOST=1 timeout 600 python3 bench.py --workload $wl --launch $mode --no-cpu-baseline --no-dice --no-roofline --steps 40 > gpurun_out/${TAG}_${wl}_${mode}.json 2> gpurun_out/${TAG}_${wl}_${mode}.err
    echo "$wl $mode rc=$? $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/${TAG}_${wl}_${mode}.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null) | $(grep 'host enqueue' gpurun_out/${TAG}_${wl}_${mode}.err | tail -1)"
    tail -3 gpurun_out/${TAG}_${wl}_${mode}.err | grep -v "host enqueue" | cut -c1-300
  done
done
for mode in graph eager; do
  BENCH_HOST=1 timeout 600 python3 bench.py --size 80 --launch $mode --no-cpu-baseline --no-dice --no-roofline --steps 40 > gpurun_out/${TAG}_mmformer80_${mode}.json 2> gpurun_out/${TAG}_mmformer80_${mode}.err
  echo "mmformer80 $mode rc=$? $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/${TAG}_mmformer80_${mode}.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null) | $(grep 'host enqueue' gpurun_out/${TAG}_mmformer80_${mode}.err | tail -1)"
done
