"""which autograd nodes / ATen ops cost the host its time?  torch.profiler (CPU activity only) over K steps at a tiny size, where
the step is host-bound.   S=32 python tools/host_nodes.py [passion] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from passion_amd import train as ptrain, optim
from passion_amd.models import mmformer
S = int(os.environ.get("S", "32"))
passion = len(sys.argv) > 1 and sys.argv[1] == "passion"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(0)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype, m.is_training = "idt", passion, torch.bfloat16, True
m.train()
opt = optim.FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4, amsgrad=True)
x, target, _ = ptrain.synthetic_batch(1, S, dev)
mask = torch.tensor([[True, True, True, True]], device=dev)


def step():
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", passion)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(K):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=60, max_name_column_width=70))
