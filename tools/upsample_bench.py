"""times the x2 trilinear upsampling (forward with fused statistics, backward) at the decoder shapes of the 128^3 step"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import ops
dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, S, C in [(4, 64, 16), (1, 64, 16), (4, 32, 32), (1, 32, 32), (4, 16, 64), (4, 8, 128)]:
    x = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16).requires_grad_(True)
    y, st = ops.upsample_with_stats(x, 2)
    gy = torch.randn_like(y)
    tf = timed(lambda: ops.upsample_with_stats(x, 2))
    def fb():
        yy, _ = ops.upsample_with_stats(x, 2)
        yy.backward(gy)
    tb = timed(fb) - tf
    mb = (y.numel() + x.numel()) * 2 / 1e6
    print(f"B={B} {S}^3 x {C} -> {2*S}^3: fwd {tf:.1f} us ({mb / tf * 1e-3 * 1e3:.0f} GB/s), bwd {tb:.1f} us ({mb / tb:.0f} GB/s), {mb:.0f} MB")
