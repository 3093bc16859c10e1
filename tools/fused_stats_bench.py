"""forward conv with and without the fused statistics epilogue, against conv + separate statistics launches (us per call)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import ops
dev = torch.device("cuda:0")
CASES = [([8], 8, 128), ([8, 8], 8, 128), ([16], 16, 64), ([16, 16], 16, 64), ([32], 32, 32), ([64], 64, 16), ([128], 128, 8)]
for segc, Cout, S in CASES:
    xs = [torch.randn(1, S, S, S, c, device=dev).to(torch.bfloat16) for c in segc]
    w = (torch.randn(Cout, sum(segc), 3, 3, 3, device=dev) * 0.1)
    b = torch.randn(Cout, device=dev)
    sts = [ops.instnorm_stats(x) for x in xs]
    segs = lambda: [ops.SegIn(x, stats=s) for x, s in zip(xs, sts)]

    def t(fn, n=100):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    with torch.no_grad():
        plain = t(lambda: ops.conv_block(segs(), w, b, S, S, S, ksize=3, stride=1, pad_mode=1))
        fused = t(lambda: ops.conv_block(segs(), w, b, S, S, S, ksize=3, stride=1, pad_mode=1, want_stats=True))
        sep = t(lambda: ops.instnorm_stats(ops.conv_block(segs(), w, b, S, S, S, ksize=3, stride=1, pad_mode=1)))
    print(f"{segc} -> {Cout} @ {S}^3: conv {plain:7.1f} us, conv+fused stats {fused:7.1f} us, conv + stats launches {sep:7.1f} us", flush=True)
