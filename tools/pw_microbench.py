"""1x1x1 forward conv alone (ops.conv_block, bf16, normalised input, fused output statistics): streaming kernel (conv_pw.hip) vs the
lean brick kernel.   python tools/pw_microbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import ops
dev = torch.device("cuda:0")
for segC, Cout, B, S in (([8], 8, 4, 128), ([16], 16, 4, 128), ([16, 16, 16, 16], 16, 2, 128), ([8, 8, 8, 8], 8, 1, 128), ([32], 32, 4, 64), ([16], 4, 2, 128)):
    xs = [(torch.randn(B, S, S, S, c, device=dev) * 2).to(torch.bfloat16) for c in segC]
    st = [ops.instnorm_stats(x) for x in xs]
    w = torch.randn(Cout, sum(segC), 1, 1, 1, device=dev) * 0.2
    b = torch.randn(Cout, device=dev)
    cache = {}
    nbytes = sum(x.numel() * 2 for x in xs) + B * S ** 3 * Cout * 2
    line = f"{segC} -> {Cout}, B={B}, {S}^3 ({nbytes / 1e6:.0f} MB): "
    for pw in (1, 0):
        ops.set_option("conv_pw", pw)
        f = lambda: ops.conv_block([ops.SegIn(x, stats=s) for x, s in zip(xs, st)], w, b, S, S, S, ksize=1, stride=1, pad_mode=0, pack_cache=cache, want_stats=True)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        line += f"{'pw' if pw else 'lean'} {ms * 1e3:.1f} us = {nbytes / ms / 1e9:.2f} TB/s   "
    print(line)
