"""the same question as tools/rfnet_cold_probe.py for mmformer (eager steps of a fresh process): do the gradients of consecutive steps on
identical inputs repeat bit for bit?  python tools/mmformer_cold_probe.py [S] [passion]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import train as ptrain
from passion_amd.models import mmformer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
passion = len(sys.argv) > 2 and sys.argv[2] == "passion"
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.compute_dtype, m.use_passion, m.is_training = "idt", torch.bfloat16, passion, True
m.eval()                     # no dropout: the step must repeat
x, t, mask = ptrain.synthetic_batch(1, S, dev)
if passion:
    mask = torch.tensor([[True, True, False, True]], device=dev)
def step():
    m.zero_grad(set_to_none=True)
    outs = m(x, mask, target=t, temp=4.0)
    st = ptrain.LossState(dev, modal_weight=torch.tensor([2.43, 1.62, 1.19, 5.09]))
    loss = ptrain.assemble_loss(outs, t, mask, "idt", passion, st)
    loss.backward()
    torch.cuda.synchronize()
    return [o.detach().clone() for o in outs], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
runs = [step() for _ in range(4)]
for i in range(1, 4):
    o_eq = all(torch.equal(a, b) for a, b in zip(runs[0][0], runs[i][0]))
    diff = [k for k in runs[0][1] if not torch.equal(runs[0][1][k], runs[i][1][k])]
    print(f"run 0 vs run {i}: outputs equal {o_eq}; {len(diff)} of {len(runs[0][1])} gradients differ {diff[:4]}")
d12 = [k for k in runs[1][1] if not torch.equal(runs[1][1][k], runs[2][1][k])]
print(f"run 1 vs run 2: {len(d12)} gradients differ")
