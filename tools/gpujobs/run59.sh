cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
timeout 900 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
done
REPS=2 STEPS=80 bash tools/ab_bench.sh "t1024:PASSION_CONV_TARGET=1024" "t2048:PASSION_CONV_TARGET=2048" "t1536:PASSION_CONV_TARGET=1536" "planes8:PASSION_MARCH_PLANES=8" "planes32:PASSION_MARCH_PLANES=32"
