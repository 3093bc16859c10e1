"""forward + backward of one 1x1x1 conv block under rocprofv3 --kernel-trace --stats: per-kernel times of the pointwise path.
   rocprofv3 --kernel-trace --stats -d DIR --output-format csv -- python3 tools/pw_bwd_trace.py C B S"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import ops
dev = torch.device("cuda:0")
C, B, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = (torch.randn(B, S, S, S, C, device=dev) * 2).to(torch.bfloat16).requires_grad_(True)
st = ops.instnorm_stats(x.detach())
w = (torch.randn(C, C, 1, 1, 1, device=dev) * 0.2).requires_grad_(True)
b = torch.randn(C, device=dev).requires_grad_(True)
gy = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
cache = {}
for _ in range(12):
    y = ops.conv_block([ops.SegIn(x, stats=st)], w, b, S, S, S, ksize=1, stride=1, pad_mode=0, pack_cache=cache)
    y.backward(gy)
    x.grad = None; w.grad = None; b.grad = None
torch.cuda.synchronize()
print(f"C={C} B={B} S={S}: tensor {B * S ** 3 * C * 2 / 1e6:.0f} MB")
