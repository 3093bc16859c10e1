cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
BENCH_HOST=1 timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/r93_$i.err | tail -1 | cut -c1-200
grep host gpurun_out/r93_$i.err
done
BENCH_HOST=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -v Warn | cut -c1-200
