#!/bin/bash
# host enqueue time per step at 32^3 (the step is host-bound there) under feature toggles; three repeats each
cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2 3; do
for e in "X=1" "PASSION_NOGC=1" "PASSION_FUSE_STREAM=0" "PASSION_WGRAD_FORK=0"; do
  echo -n "$e: "; env $e S=32 python tools/host_profile.py plain 40 2>/dev/null | grep "host enqueue"
done
done
