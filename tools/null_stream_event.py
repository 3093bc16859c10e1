"""does an event recorded on torch.cuda.ExternalStream(0) (the raw handle of the default stream) order a side stream behind the
default stream's pending work?  (parallel.GradReducer._launch relies on it)"""
import threading, torch
dev = torch.device("cuda:0")
s = torch.cuda.Stream()
a = torch.zeros(1 << 20, device=dev)
big = torch.randn(64 << 20, device=dev)
for mode in ("ExternalStream(0)", "default_stream()", "current_stream()"):
    bad = 0
    for it in range(1, 201):
        for _ in range(6):
            big.mul_(1.0000001)               # keep the default stream busy
        a.fill_(float(it))
        ev = torch.cuda.Event()
        src = {"ExternalStream(0)": torch.cuda.ExternalStream(0), "default_stream()": torch.cuda.default_stream(),
               "current_stream()": torch.cuda.current_stream()}[mode]
        ev.record(src)
        with torch.cuda.stream(s):
            s.wait_event(ev)
            b = a.clone()
        torch.cuda.synchronize()
        bad += int(float(b[0]) != float(it) or float(b[-1]) != float(it))
    print(f"{mode:22s}: {bad} of 200 copies on the side stream ran ahead of the default stream's fill")
print("raw handle of the default stream:", hex(torch.cuda.default_stream().cuda_stream), "current:", hex(torch.cuda.current_stream().cuda_stream))

# and the thing the whole library relies on: a kernel the library launches on the raw handle of torch's current (default) stream is
# ordered behind torch's own pending work on that stream, and torch's next op behind it
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passion_amd import ops
g = torch.zeros(256, 512, device=dev)
bad = bad2 = 0
for it in range(1, 201):
    for _ in range(6):
        big.mul_(1.0000001)
    g.fill_(float(it))                       # torch op on the default stream
    out = ops.colsum(g)                      # library kernel on raw handle 0
    back = out * 2.0                         # torch op reading the library's result
    torch.cuda.synchronize()
    bad += int(float(out[0]) != 256.0 * it)
    bad2 += int(float(back[-1]) != 512.0 * it)
print(f"library kernel after torch op on the default stream: {bad} of 200 wrong; torch op after the library kernel: {bad2} of 200 wrong")
