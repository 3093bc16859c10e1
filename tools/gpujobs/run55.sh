cd $GRAFT_REPO_ROOT
R=$PWD
for wl in mmformer128; do
mkdir -p gpurun_out/prof_$wl
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$wl -o st --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --workload $wl > $R/gpurun_out/prof_$wl/bench.json 2> $R/gpurun_out/prof_$wl/bench.err)
find gpurun_out/prof_$wl -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02i_${wl}_kernel_stats.csv \;
rm -rf gpurun_out/prof_$wl
done
timeout 900 python tools/host_profile.py plain 20 2>&1 | grep -v amdgpu > gpurun_out/host_profile_plain.txt
head -12 gpurun_out/host_profile_plain.txt
