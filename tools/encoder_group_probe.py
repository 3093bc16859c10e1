"""What would one launch per layer for the four modality encoders buy?  (a) the four encoders, one sample each, on four
streams (the product path); (b) ONE encoder over a batch of four on one stream -- the kernel work of a grouped launch
(per-sample weights would not change it).  Forward + backward, bf16, 128^3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
encs = [mmformer.Encoder().to(dev) for _ in range(4)]
x1 = torch.randn(1, 4, S, S, S, device=dev).to(torch.bfloat16)
x4 = torch.randn(4, 4, S, S, S, device=dev).to(torch.bfloat16)
streams = [torch.cuda.Stream() for _ in range(4)]
main = torch.cuda.current_stream()


def four_streams():
    outs = []
    for i in range(4):
        streams[i].wait_stream(main)
        with torch.cuda.stream(streams[i]):
            f = encs[i](x1, i, None, torch.bfloat16)
            outs.append(sum(t.t.float().mean() for t in f))
    for s in streams:
        main.wait_stream(s)
    sum(outs).backward()


def one_batch4():
    f = encs[0](x4, 0, None, torch.bfloat16)
    sum(t.t.float().mean() for t in f).backward()


def four_serial():
    outs = []
    for i in range(4):
        f = encs[i](x1, i, None, torch.bfloat16)
        outs.append(sum(t.t.float().mean() for t in f))
    sum(outs).backward()


for name, fn in (("four encoders on four streams", four_streams), ("four encoders, one stream", four_serial), ("one encoder, batch of four", one_batch4)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tw = time.perf_counter() - t0
    print(f"{name:34s}: host {1e3 * th / n:6.2f} ms, wall {1e3 * tw / n:6.2f} ms", flush=True)
