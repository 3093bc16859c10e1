"""host cost per call of a few tiny ops (GPU work negligible): where does the per-launch time go?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import ops, _lib as L
dev = torch.device("cuda:0")
x = torch.randn(1, 8, 8, 8, 8, device=dev).to(torch.bfloat16)
a = torch.randn(1024, device=dev)
out = torch.empty(1, 8, 2, device=dev)

def bench(name, fn, n=3000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:40s} host {1e6*(t1-t0)/n:7.2f} us/call   incl. drain {1e6*(t2-t0)/n:7.2f} us/call", flush=True)

bench("torch add (1 launch)", lambda: a + a)
bench("torch.empty", lambda: torch.empty(16, device=dev))
bench("instnorm_stats (2 launches, 2 empties)", lambda: ops.instnorm_stats(x))
bench("instnorm_stats out= (2 launches, 1 empty)", lambda: ops.instnorm_stats(x, out=out))
lib = L.load()
from ctypes import c_void_p
scr = torch.empty(lib.passion_instnorm_scratch_doubles(1, 512, 8), dtype=torch.float64, device=dev)
args = (c_void_p(x.data_ptr()), 1, 1, 512, 8, 512 * 8, 1e-5, c_void_p(scr.data_ptr()), c_void_p(out.data_ptr()))
bench("raw ctypes passion_instnorm_stats", lambda: lib.passion_instnorm_stats(*args, L.stream_ptr()))
s2 = torch.cuda.Stream()
def two_streams():
    lib.passion_instnorm_stats(*args, c_void_p(0))
    lib.passion_instnorm_stats(*args, c_void_p(s2.cuda_stream))
bench("raw ctypes, alternating 2 streams (4 launches)", two_streams)
w = torch.randn(8, 8, 3, 3, 3, device=dev, requires_grad=True); b = torch.zeros(8, device=dev, requires_grad=True)
xs = x.clone().requires_grad_(True)
st = ops.instnorm_stats(x)
cache = {}
def conv_f():
    return ops.conv_block([ops.SegIn(xs, stats=st)], w, b, 8, 8, 8, pack_cache=cache)
bench("conv_block fwd (1 launch)", conv_f)
y = conv_f(); gy = torch.randn_like(y)
def conv_fb():
    y = conv_f(); y.backward(gy)
bench("conv_block fwd+bwd", conv_fb, 1000)
