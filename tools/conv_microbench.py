"""single-layer micro-benchmark of the fused conv kernels (for rocprofv3 --pmc passes and A/B timing)
usage: conv_microbench.py MODE CIN COUT S [K] [STRIDE] [ITERS]      MODE = fwd | bwd"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import ops, _lib as L

mode, cin, cout, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 50
dev = torch.device("cuda:0")
torch.manual_seed(0)
segs_c = [8] * (cin // 8) if cin > 8 and cin <= 32 and not os.environ.get('MB_ONESEG') else [cin]
NB = int(os.environ.get('MB_BATCH', '1'))
xs = [torch.randn(NB, S, S, S, c, device=dev).to(torch.bfloat16).requires_grad_(mode == "bwd") for c in segs_c]
w = (torch.randn(cout, cin, k, k, k, device=dev) / (cin * k ** 3) ** 0.5).requires_grad_(True)
b = torch.zeros(cout, device=dev, requires_grad=True)
stats = [ops.instnorm_stats(x.detach()) for x in xs]
cache = {}

def run():
    segs = [ops.SegIn(x, stats=st) for x, st in zip(xs, stats)]
    return ops.conv_block(segs, w, b, S, S, S, ksize=k, stride=stride, pad_mode=L.PAD_REFLECT, pack_cache=cache)

y = run()
gy = torch.randn_like(y)
for _ in range(3):
    y = run()
    if mode == "bwd":
        y.backward(gy)
torch.cuda.synchronize()
ops.prof_enable(15 | 64)
t0 = time.time()
for _ in range(iters):
    y = run()
    if mode == "bwd":
        y.backward(gy)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
ops.prof_enable(0)
recs = ops.prof_read()
names = {0: "fwd_k3", 1: "fwd_k1", 2: "dgrad", 3: "wgrad", 6: "finish"}
So = S // stride
alg = NB * (cin * S ** 3 + cout * So ** 3) * 2
print(f"{mode} cin={cin} (segments {segs_c}) cout={cout} S={S} k={k} stride={stride}: wall {dt*1e6:.1f} us/iter, algorithmic {alg/1e6:.1f} MB")
for kind in (0, 1, 2, 3, 6):
    r = [x for x in recs if x[0] == kind]
    if r:
        ms = sum(x[1] for x in r) / len(r)
        print(f"  {names[kind]}: {ms*1e3:.1f} us  {r[0][2]/ms/1e6:.0f} GB/s  {r[0][3]/ms/1e9:.1f} TFLOP/s  {r[0][4]}")
