#!/bin/bash
# A/B of environment switches on the graph-replayed bench.  usage: tools/ab_graph.sh "LABEL:ENV=.. ENV=.." ... [-- bench args]
cfgs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do cfgs+=("$1"); shift; done; [ "$1" == "--" ] && shift
REPS=${REPS:-2}
for r in $(seq 1 $REPS); do
  for c in "${cfgs[@]}"; do
    label=${c%%:*}; envs=${c#*:}
    env $envs timeout 900 python3 bench.py --no-cpu-baseline --no-roofline --no-dice --steps ${STEPS:-100} "$@" > /tmp/ab_out.txt 2> /tmp/ab_err.txt
    v=$(tail -1 /tmp/ab_out.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$label rep$r: $v"
  done
done
