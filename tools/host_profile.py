"""cProfile of the host side of one training step (tiny volume: the GPU is never the bottleneck): where do the ~23 ms go?"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
from passion_amd import train as ptrain
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", False, torch.bfloat16
m.train()
opt = ptrain.make_optimizer(m)
state = ptrain.LossState(dev)
x, target, mask = ptrain.synthetic_batch(1, S, dev)
for _ in range(8):
    ptrain.train_step(m, opt, x, mask, target, 4.0, state, None)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
with torch.autograd.set_multithreading_enabled(False):      # backward in this thread, so that the profiler sees it
    for _ in range(10):
        ptrain.train_step(m, opt, x, mask, target, 4.0, state, None)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(70)
    print(s.getvalue()[:14000])
