"""token GEMMs of the transformers' feed-forward layers (reference mmformer.py:270-276: Linear(512, 4096), GELU, Linear(4096, 512)) on hipBLASLt /
rocBLAS: which call form gets which kernel.  M = tokens (2048 plain, 10240 = 5 virtual samples under PASSION).  python tools/gemm_ffn_bench.py"""
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for M in (2048, 10240):
        for K, N in ((512, 4096), (4096, 512), (512, 1536), (512, 512)):
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
            wt = w.t().contiguous()
            b = torch.randn(N, device=dev, dtype=torch.bfloat16)
            g = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
            gf = 2.0 * M * K * N / 1e9
            res = {
                "linear+bias": t(lambda: F.linear(x, w, b)),
                "linear": t(lambda: F.linear(x, w)),
                "mm(x, wt)": t(lambda: torch.mm(x, wt)),
                "addmm(b, x, wt)": t(lambda: torch.addmm(b, x, wt)),
                "dgrad mm(g, w)": t(lambda: torch.mm(g, w)),
                "wgrad mm(g.t, x)": t(lambda: torch.mm(g.t(), x)),
                "wgrad f32 out": t(lambda: torch.mm(g.t(), x, out_dtype=torch.float32)),
            }
            print(f"{lib:9s} M={M:6d} K={K:5d} N={N:5d} ({gf:6.1f} GFLOP): " + "  ".join(f"{k} {v:6.1f} us ({gf / v * 1e3:5.0f} TF)" for k, v in res.items()))
