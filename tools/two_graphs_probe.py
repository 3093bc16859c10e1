"""Do two recorded training steps, launched on two streams, overlap on the GPU?  Two model replicas, two GraphedSteps: time K
iterations of one alone, then of both (launched back to back on their own streams each iteration).  If the pair costs clearly less
than 2 x one, graph launches on different streams do run side by side, and the serial look of ONE recorded step (80 % of its span
with a single kernel in flight) is the graph executor's, not the hardware's.  python tools/two_graphs_probe.py [S] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import train as ptrain, graph as pgraph
from passion_amd.models import mmformer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
def make(seed):
    torch.manual_seed(seed)
    m = mmformer.Model().to(dev)
    m.mask_type, m.use_passion, m.compute_dtype, m.is_training = "idt", False, torch.bfloat16, True
    m.train()
    opt = ptrain.make_optimizer(m)
    return pgraph.GraphedStep(m, opt, temp=4.0)
x, t, mask = ptrain.synthetic_batch(1, S, dev)
ga, gb = make(1), make(2)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
xa, ta, xb, tb = x.clone(), t.clone(), x.clone(), t.clone()
def run(gs, st, xx, tt):
    with torch.cuda.stream(st):
        gs(xx, mask, tt)
for _ in range(6):
    run(ga, sa, xa, ta); run(gb, sb, xb, tb)
torch.cuda.synchronize()
ia, ib = ga.input_buffers(), gb.input_buffers()
def timeit(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
one = timeit(lambda: run(ga, sa, ia[0], ia[2]))
two = timeit(lambda: (run(ga, sa, ia[0], ia[2]), run(gb, sb, ib[0], ib[2])))
print(f"one recorded step {one:.2f} ms; two on two streams {two:.2f} ms per pair = {two / one:.2f} x one")
