"""forward + backward wall time of 1x1x1 conv blocks: finish-free backward (option conv_pwfinal) vs dgrad + finish.
   python tools/pw_bwd_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import ops
dev = torch.device("cuda:0")
for C, B, S in ((16, 4, 128), (8, 4, 128), (8, 1, 128), (32, 4, 64), (16, 1, 128)):
    x = (torch.randn(B, S, S, S, C, device=dev) * 2).to(torch.bfloat16).requires_grad_(True)
    st = ops.instnorm_stats(x.detach())
    w = (torch.randn(C, C, 1, 1, 1, device=dev) * 0.2).requires_grad_(True)
    b = torch.randn(C, device=dev).requires_grad_(True)
    gy = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
    cache = {}
    line = f"C={C} B={B} {S}^3: "
    for fin in (1, 0):
        ops.set_option("conv_pwfinal", fin)
        def f():
            y = ops.conv_block([ops.SegIn(x, stats=st)], w, b, S, S, S, ksize=1, stride=1, pad_mode=0, pack_cache=cache)
            y.backward(gy)
            x.grad = None; w.grad = None; b.grad = None
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            f()
        e1.record()
        torch.cuda.synchronize()
        line += f"{'finish-free' if fin else 'dgrad+finish'} {e0.elapsed_time(e1) / 30 * 1e3:.0f} us   "
    print(line)
