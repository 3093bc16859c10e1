#!/bin/bash
# A/B of environment switches on bench.py, interleaved repeats.   usage: tools/ab_bench.sh "LABEL_A:ENV=.. ENV=.." "LABEL_B:..." [-- bench args]
# (each run its own process: the switches are read at import)
cd ${GRAFT_REPO_ROOT:-.}
cfgs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do cfgs+=("$1"); shift; done; [ "$1" == "--" ] && shift
REPS=${REPS:-3}
for r in $(seq 1 $REPS); do
  for c in "${cfgs[@]}"; do
    label=${c%%:*}; envs=${c#*:}
    env $envs timeout 900 python bench.py --no-cpu-baseline --no-roofline --steps ${STEPS:-150} "$@" > /tmp/ab_out.txt 2> /tmp/ab_err.txt
    v=$(tail -1 /tmp/ab_out.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    h=$(grep "host enqueue" /tmp/ab_err.txt | tail -1 | sed 's/\[bench\] //')
    echo "$label rep$r: $v | $h"
  done
done
