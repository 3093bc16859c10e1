"""which ATen ops (the torch glue between the library's kernels) a training step still launches: count, device time, shapes.
    python tools/torch_ops_table.py [passion 0/1] [mmformer | rfnet | m2ftrans] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.profiler import profile, ProfilerActivity
from passion_amd import train as ptrain
from passion_amd.models import mmformer

passion = len(sys.argv) > 1 and sys.argv[1] == "1"
backbone = sys.argv[2] if len(sys.argv) > 2 else "mmformer"      # mmformer | rfnet | m2ftrans
dev = torch.device("cuda:0")
mmformer.patch_size = 8
torch.manual_seed(1037)
if backbone == "rfnet":
    from passion_amd.models import rfnet
    m = rfnet.Model(num_cls=4).to(dev)
elif backbone == "m2ftrans":
    from passion_amd.models import m2ftrans
    m2ftrans.patch_size = 8
    m = m2ftrans.Model(num_cls=4).to(dev)
else:
    m = mmformer.Model(num_cls=4).to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, torch.bfloat16
m.train()
opt = ptrain.make_optimizer(m)
state = ptrain.LossState(dev)
x, target, mask = ptrain.synthetic_batch(int(sys.argv[3]) if len(sys.argv) > 3 else 1, 128, dev)
for _ in range(4):
    ptrain.train_step(m, opt, x, mask, target.view(target.shape), 4.0, state)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ptrain.train_step(m, opt, x, mask, target.view(target.shape), 4.0, state)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten ops with device time: {sum(r[1] for r in rows)} calls, {tot / 1e3:.3f} ms")
for dt, n, k, sh in rows[:60]:
    print(f"{dt / 1e3:8.3f} ms n={n:3d} {k:28s} {sh}")
