cd $GRAFT_REPO_ROOT
R=$PWD
wl=mmformer128
mkdir -p gpurun_out/prof_$wl
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$wl -o st --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --workload $wl > $R/gpurun_out/prof_$wl/bench.json 2> $R/gpurun_out/prof_$wl/bench.err)
find gpurun_out/prof_$wl -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02l_${wl}_kernel_stats.csv \;
rm -rf gpurun_out/prof_$wl
timeout 900 python tools/determinism_check.py 64 passion train 40 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-150 | awk 'NR<=3 || /pairs differ/'
timeout 900 python tools/determinism_check.py 64 plain train 40 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-150 | awk 'NR<=3 || /pairs differ/'
REPS=2 STEPS=60 bash tools/ab_bench.sh "passion:X=1" -- --workload mmformer128_passion
