for i in 1 2 3 4; do
    r=$(python -m pytest tests/test_fullsize_gpu.py -q -x -k rfnet_128 2>&1 | tail -1)
    echo "rfnet test : $r"
done > gpurun_out/e1.txt 2>&1
python -m pytest tests/test_fullsize_gpu.py tests/test_rfnet_gpu.py -q 2>&1 | tail -2 >> gpurun_out/e1.txt
