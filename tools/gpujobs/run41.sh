cd $GRAFT_REPO_ROOT
REPS=4 STEPS=100 bash tools/ab_bench.sh "fused:PASSION_FUSE_STATS=1" "separate:PASSION_FUSE_STATS=0"
nproc; cat /proc/cpuinfo | grep "model name" | head -1; uptime
