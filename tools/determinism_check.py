"""run the same bf16 training step twice from identical state: which gradients repeat bit for bit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import train as ptrain
from passion_amd.models import mmformer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
passion = (sys.argv[2] == "passion") if len(sys.argv) > 2 else True
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, torch.bfloat16
m.eval()
x, target, _ = ptrain.synthetic_batch(1, S, dev)
mask = torch.tensor([[True, True, False, True]], device=dev)
runs = []
for r in range(3):
    for p in m.parameters():
        p.grad = None
    m.is_training = True
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", passion)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((float(loss), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, [o.detach().clone() for o in outs]))
print("losses", [r[0] for r in runs])
for i, o in enumerate(runs[0][2]):
    print("output", i, "bit-equal:", all(torch.equal(o, r[2][i]) for r in runs[1:]))
bad = [k for k in runs[0][1] if not all(torch.equal(runs[0][1][k], r[1][k]) for r in runs[1:])]
print(f"{len(bad)} of {len(runs[0][1])} gradient tensors differ between runs")
for k in bad[:40]:
    d = max(float((runs[0][1][k] - r[1][k]).abs().max()) for r in runs[1:])
    print(f"   {k:60s} max diff {d:.3e} / {float(runs[0][1][k].abs().max()):.3e}")
