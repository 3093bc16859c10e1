cd _wt
for i in 1 2 3 4 5 6; do
    r=$(python -m pytest tests/test_fullsize_gpu.py -q -x -k rfnet_128 2>&1 | tail -1)
    echo "round5 tree : $r"
done > ../gpurun_out/e1.txt 2>&1
