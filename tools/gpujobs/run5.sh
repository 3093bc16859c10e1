set -x
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh gpurun_out/pmc_march_fwd -- python3 tools/conv_microbench.py fwd 8 8 128 3 1 10 2>&1 | tail -40
