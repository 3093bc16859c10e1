cd $GRAFT_REPO_ROOT
export MB_BATCH=4
bash tools/pmc_run.sh gpurun_out/pmc_wgrad_march -- python3 tools/conv_microbench.py bwd 8 8 128 3 1 10 > gpurun_out/r02n_pmc_wgrad_march.txt 2>&1
PASSION_WGRAD_MARCH=0 bash tools/pmc_run.sh gpurun_out/pmc_wgrad_lean -- python3 tools/conv_microbench.py bwd 8 8 128 3 1 10 > gpurun_out/r02n_pmc_wgrad_lean.txt 2>&1
grep -A 22 "wgrad_march" gpurun_out/r02n_pmc_wgrad_march.txt | head -30
grep -A 22 "wgrad_lean_kernel" gpurun_out/r02n_pmc_wgrad_lean.txt | head -30
rm -rf gpurun_out/pmc_wgrad_march gpurun_out/pmc_wgrad_lean
