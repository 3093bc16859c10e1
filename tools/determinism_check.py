"""run the same bf16 training step N times from identical state: which outputs / gradients repeat bit for bit?
usage: determinism_check.py S plain|passion eval|train [runs] [synthetic|blobs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from passion_amd import train as ptrain
from passion_amd.models import mmformer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
passion = (sys.argv[2] == "passion") if len(sys.argv) > 2 else True
train = len(sys.argv) > 3 and sys.argv[3] == "train"
N = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
if len(sys.argv) > 5 and sys.argv[5] == "blobs":
    from _util import build_product_model, synth
    m = build_product_model(S, small=False).to(dev)
    x, target = synth(1, S, blobs=True)
    x, target = x.to(dev), target.to(dev)
else:
    mmformer.patch_size = S // 16
    torch.manual_seed(1037)
    m = mmformer.Model().to(dev)
    x, target, _ = ptrain.synthetic_batch(1, S, dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, torch.bfloat16
m.eval()
if train:
    m.train()
mask = torch.tensor([[True, True, False, True]], device=dev)
runs = []
for r in range(N):
    torch.manual_seed(11)
    for p in m.parameters():
        p.grad = None
    m.is_training = True
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", passion)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, [o.detach().clone() for o in outs]))
print("losses", sorted(set(r[0] for r in runs)))
for i, o in enumerate(runs[0][2]):
    if not all(torch.equal(o, r[2][i]) for r in runs[1:]):
        print("output", i, "differs")
nbad = 0
for k in runs[0][1]:
    for ri, r in enumerate(runs[1:]):
        a, b = runs[0][1][k].flatten(), r[1][k].flatten()
        if not torch.equal(a, b):
            idx = (a != b).nonzero().flatten()
            nbad += 1
            print(f"  run {ri + 1}: {k} {tuple(runs[0][1][k].shape)}: {idx.numel()} elements differ, first {idx[:6].tolist()} last {idx[-1].item()}, "
                  f"max diff {float((a - b).abs().max()):.3e} of {float(a.abs().max()):.3e}; e.g. {float(a[idx[0]]):.6e} vs {float(b[idx[0]]):.6e}")
print(f"{nbad} (tensor, run) pairs differ from run 0, {N} runs")
