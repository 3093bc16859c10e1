#!/usr/bin/env python
"""Attention core alone (ops.attention on random qkv): timing with hipEvents, and the target of tools/pmc_attention.sh.
   python tools/attn_microbench.py [--B 5 --N 2048 --drop 0.1 --iters 20 --no-bwd]   (defaults = the inter-modal call of a PASSION step)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from passion_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=5)
ap.add_argument("--N", type=int, default=2048)
ap.add_argument("--H", type=int, default=8)
ap.add_argument("--Dh", type=int, default=64)
ap.add_argument("--drop", type=float, default=0.1)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--no-bwd", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
torch.manual_seed(0)
qkv = (torch.randn(a.B, a.N, 3 * a.H * a.Dh, device=dev) * 1.0).to(dt).requires_grad_(True)
go = torch.randn(a.B, a.N, a.H * a.Dh, device=dev).to(dt)
scale = a.Dh ** -0.5


def run():
    out = ops.attention(qkv, a.H, scale, a.drop)
    if not a.no_bwd:
        qkv.grad = None
        out.backward(go)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
fl = 4.0 * a.B * a.H * a.N * a.N * a.Dh * (1.0 if a.no_bwd else 3.5)      # 4*B*N^2*512 forward, backward x2.5
print(f"attention B={a.B} N={a.N} H={a.H} Dh={a.Dh} {a.dtype} drop={a.drop} {'fwd' if a.no_bwd else 'fwd+bwd'}: "
      f"{ms:.3f} ms/iter, {fl / ms / 1e9:.1f} TFLOP/s algorithmic ({fl / 1e9:.1f} GFLOP)")
