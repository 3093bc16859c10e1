R=$PWD
rm -rf /tmp/trf; mkdir -p /tmp/trf
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/trf -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-dice --no-roofline --steps 8 --warmup 2 > /tmp/trf/bench.log 2>&1
cd $R
python3 tools/trace_torch.py $(find /tmp/trf -name "*kernel_trace.csv" | head -1)
