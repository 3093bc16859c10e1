"""x2 trilinear upsampling (ops.upsample / upsample_with_stats) forward + backward at tiny volumes against F.interpolate in float64, memory
pre-filled with garbage between trials (a gradient that depends on what was in the allocator's blocks before would show here)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from passion_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
for shape in [(1, 2, 2, 2, 64), (1, 4, 4, 4, 32), (1, 8, 8, 8, 16), (5, 4, 4, 4, 32), (1, 4, 4, 4, 64), (1, 2, 2, 2, 256)]:
    for dtype in (torch.float32, torch.bfloat16):
        worst = [0.0, 0.0]
        for trial in range(6):
            junk = [torch.full((1 << 20,), float("nan") if trial % 2 else 1e30, device=dev) for _ in range(4)]
            del junk
            B, D, H, W, C = shape
            x = torch.randn(B, D, H, W, C, generator=g).to(dtype)
            gy = torch.randn(B, 2 * D, 2 * H, 2 * W, C, generator=g).to(dtype)
            xr = x.double().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
            yr = F.interpolate(xr, scale_factor=2, mode="trilinear", align_corners=True)
            yr.backward(gy.double().permute(0, 4, 1, 2, 3))
            for with_stats in (False, True):
                xd = x.to(dev).requires_grad_(True)
                y = ops.upsample_with_stats(xd, 2)[0] if with_stats else ops.upsample(xd, 2)
                y.backward(gy.to(dev))
                ef = float((y.detach().double().cpu().permute(0, 4, 1, 2, 3) - yr.detach()).norm() / yr.detach().norm())
                eb = float((xd.grad.double().cpu().permute(0, 4, 1, 2, 3) - xr.grad).norm() / xr.grad.norm())
                worst = [max(worst[0], ef), max(worst[1], eb)]
        print(f"{shape} {str(dtype)[6:]:9s} worst fwd {worst[0]:.1e} bwd {worst[1]:.1e}")
