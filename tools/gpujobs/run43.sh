cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch
a=torch.randn(64,32,device='cuda',dtype=torch.bfloat16); b=torch.randn(32,16,device='cuda',dtype=torch.bfloat16)
try:
    print("mm out_dtype on GPU:", torch.mm(a,b,out_dtype=torch.float32).dtype)
except Exception as e:
    print("mm out_dtype fails:", str(e)[:200])
PY
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_benchsize_gpu.py tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -3
REPS=3 STEPS=100 bash tools/ab_bench.sh "hipblaslt:X=1" "rocblas:TORCH_BLAS_PREFER_HIPBLASLT=0"
