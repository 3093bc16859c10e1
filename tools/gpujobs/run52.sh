cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -8
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_fullsize_gpu.py 2>&1 | tail -4
for i in 1 2 3; do
timeout 900 python tools/determinism_check.py 64 plain train 60 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-170 > gpurun_out/det_$i.txt
head -8 gpurun_out/det_$i.txt; tail -1 gpurun_out/det_$i.txt
done
