cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
timeout 900 python tools/determinism_check.py 64 plain train 60 blobs 2>&1 | grep -v "amdgpu\|Warning\|runs.append\|Consider" | cut -c1-170 > gpurun_out/det_$i.txt
head -14 gpurun_out/det_$i.txt; tail -1 gpurun_out/det_$i.txt
done
