cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh "fused:PASSION_FUSE_STATS=1" "separate:PASSION_FUSE_STATS=0"
REPS=2 bash tools/ab_bench.sh "passion_fused:PASSION_FUSE_STATS=1" "passion_separate:PASSION_FUSE_STATS=0" -- --workload mmformer128_passion
