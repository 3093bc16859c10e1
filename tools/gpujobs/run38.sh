cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/prof_tl
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 3 > $R/gpurun_out/prof_tl/bench.json 2> $R/gpurun_out/prof_tl/bench.err)
f=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
head -1 $f
python tools/timeline.py $f
python tools/timeline.py $f 10
rm -rf gpurun_out/prof_tl
