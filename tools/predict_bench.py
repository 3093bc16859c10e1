"""N1 measurement: sliding-window inference of one full 240x240x155 volume (reference predict.py:182-218, 128^3 windows)
on the product path, bf16, forward only.  usage: predict_bench.py [patch] [windows_per_batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
from passion_amd.utils import predict as pp

patch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
wpb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
mmformer.patch_size = patch // 16
torch.manual_seed(1037)
m = mmformer.Model(num_cls=4).to(dev)
m.mask_type, m.compute_dtype = "idt", torch.bfloat16
m.eval()
x = torch.randn(1, 4, 240, 240, 155, generator=torch.Generator().manual_seed(7)).to(dev)
mask = torch.tensor([[True, False, True, True]], device=dev)
nwin = len(pp.window_starts(240, patch)) ** 2 * len(pp.window_starts(155, patch))
with torch.no_grad():
    for _ in range(2):
        pp.sliding_window_predict(m, x, mask, patch, windows_per_batch=wpb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        prob, seg = pp.sliding_window_predict(m, x, mask, patch, windows_per_batch=wpb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print(f"240x240x155 volume, {nwin} windows of {patch}^3 ({wpb} per batch), one mask: {1e3 * dt:.1f} ms per volume = "
      f"{nwin / dt:.1f} windows/s; the reference evaluates 15 masks per test volume: {15 * dt:.2f} s per volume")
