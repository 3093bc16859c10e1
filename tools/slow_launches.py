"""one plain (or PASSION) 128^3 step with every conv launch timed: the launches above a duration, with their algorithmic bytes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import ops, train as ptrain, optim
from passion_amd.models import mmformer
passion = len(sys.argv) > 1 and sys.argv[1] == "passion"
dev = torch.device("cuda:0")
mmformer.patch_size = 8
mmformer.ENC_STREAMS = False
torch.manual_seed(0)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype, m.is_training = "idt", passion, torch.bfloat16, True
m.train()
opt = optim.FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4, amsgrad=True)
x, target, _ = ptrain.synthetic_batch(1, 128, dev)
mask = torch.tensor([[True, True, True, True]], device=dev)


def step():
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", passion)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
ops.prof_read()
ops.prof_enable(0x7f)
step()
torch.cuda.synchronize()
ops.prof_enable(0)
recs = [r for r in ops.prof_read() if r[1] >= 0]
tot = sum(r[1] for r in recs)
print(f"{len(recs)} timed launches, {tot:.2f} ms")
for r in sorted(recs, key=lambda r: -r[1])[:40]:
    print(f"{r[1]*1e3:8.1f} us  {r[2]/1e6:8.1f} MB  {r[2]/r[1]/1e9*1e3/8000:6.3f} of 8 TB/s   kind {r[0]}  {r[4]}")
