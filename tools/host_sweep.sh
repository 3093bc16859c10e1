cd ${GRAFT_REPO_ROOT:-.}
for e in "X=1" "PASSION_EARLY_STEP=0" "PASSION_LOSS_STREAMS=0" "PASSION_FUSE_STREAM=0" "PASSION_CONV_PWFINAL=0" "PASSION_WGRAD_FORK=0"; do
  echo -n "$e: "; env $e S=32 python tools/host_profile.py plain 30 2>/dev/null | grep "host enqueue"
done
