import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from passion_amd import train as ptrain
from passion_amd.models import mmformer, rfnet
dev = torch.device("cuda:0")
torch.manual_seed(1037)
m = rfnet.Model(num_cls=4).to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", True, torch.bfloat16
m.train()
opt = ptrain.make_optimizer(m)
state = ptrain.LossState(dev)
x, target, mask = ptrain.synthetic_batch(2, 128, dev)
for _ in range(2):
    ptrain.train_step(m, opt, x, mask, target.view(target.shape), 4.0, state)
log = collections.Counter()
orig = torch.cat
def cat(ts, dim=0, **kw):
    ts = list(ts)
    n = sum(t.numel() * t.element_size() for t in ts)
    if n > 1 << 20:
        fr = traceback.extract_stack(limit=4)[:-1]
        log[(tuple(tuple(t.shape) for t in ts), dim, " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
    return orig(ts, dim, **kw)
torch.cat = cat
ptrain.train_step(m, opt, x, mask, target.view(target.shape), 4.0, state)
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -sum(__import__("math").prod(s) for s in kv[0][0])):
    print(v, k)
