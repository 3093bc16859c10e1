"""where the D2D copies / torch glue kernels of a training step come from: one step under torch.profiler with Python stacks, ATen ops that
launch `copy` / `add` / `sum` kernels grouped by call site.  usage: copy_sites.py [rfnet|mmformer|m2ftrans] [passion]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from passion_amd import train as ptrain

which = sys.argv[1] if len(sys.argv) > 1 else "rfnet"
passion = len(sys.argv) > 2 and sys.argv[2] == "passion"
dev = torch.device("cuda:0")
S = 128
torch.manual_seed(1037)
if which == "rfnet":
    from passion_amd.models import rfnet
    model, batch = rfnet.Model(num_cls=4).to(dev), 2
elif which == "m2ftrans":
    from passion_amd.models import m2ftrans
    m2ftrans.patch_size = S // 16
    model, batch = m2ftrans.Model(num_cls=4).to(dev), 1
else:
    from passion_amd.models import mmformer
    mmformer.patch_size = S // 16
    model, batch = mmformer.Model(num_cls=4).to(dev), 1
model.mask_type, model.use_passion, model.compute_dtype = "idt", passion, torch.bfloat16
model.train()
opt = ptrain.make_optimizer(model)
state = ptrain.LossState(dev)
x, target, mask = ptrain.synthetic_batch(batch, S, dev, seed=7)


def step():
    return ptrain.train_step(model, opt, x, mask, target.view(target.shape), 4.0, state, None)


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
sites = {}
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::clone", "aten::contiguous", "aten::cat", "aten::mul"):
        if ev.name == "aten::copy_" and ev.input_shapes and len(ev.input_shapes) > 1 and ev.input_shapes[0] != ev.input_shapes[1]:
            pass
        st = [f for f in (ev.stack or []) if "passion_amd" in f or "autograd" in f][:3]
        key = (ev.name, str(ev.input_shapes)[:80], " <- ".join(s.split("/")[-1] for s in st))
        sites[key] = sites.get(key, 0) + 1
for (name, shp, st), n in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n:4d}x {name:16s} {shp:80s} {st}")
