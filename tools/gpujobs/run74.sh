cd $GRAFT_REPO_ROOT
timeout 900 python tools/host_profile.py plain 20 2>&1 | grep -v amdgpu > gpurun_out/host_profile_plain.txt; head -3 gpurun_out/host_profile_plain.txt
timeout 900 python tools/host_profile.py passion 20 2>&1 | grep -v amdgpu > gpurun_out/host_profile_passion.txt; head -3 gpurun_out/host_profile_passion.txt
