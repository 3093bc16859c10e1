cd $GRAFT_REPO_ROOT
REPS=2 STEPS=100 bash tools/ab_bench.sh "lazyfork:PASSION_WGRAD_FORK=2" "nofork:PASSION_WGRAD_FORK=0"
