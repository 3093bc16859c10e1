#!/bin/bash
# A/B of the lean kernels' row-tile rule (PASSION_LEAN_MT_TILES): whole-step rate, alternating, then per-kernel tables
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-dice --no-roofline --steps 40 --warmup 10 >/dev/null 2>&1
for i in 1 2 3 4; do
for mt in ${MTS:-512 0}; do
  echo "MT_TILES=$mt: $(PASSION_LEAN_MT_TILES=$mt python bench.py --no-cpu-baseline --no-dice --no-roofline --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
done
for mt in ${MTS:-512 0}; do PASSION_LEAN_MT_TILES=$mt bash tools/kernel_ab.sh mt$mt > /dev/null 2>&1; done
