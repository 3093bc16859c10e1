"""per-parameter gradient difference product vs oracle-on-GPU (f32) for one masks15_train.npz row, teacher-only loss (fuse + prm)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from _util import gold, synth, small_weights
from test_masks15_gpu import _build
from oracle import mmformer_ref
from oracle import criterions_ref as C
from passion_amd.utils import criterions
g = gold("masks15_train.npz"); S = int(g["S"]); dev = torch.device("cuda:0")
row = int(sys.argv[1]) if len(sys.argv) > 1 else 1
passion = (sys.argv[2] != "plain") if len(sys.argv) > 2 else True
x, target = synth(1, S, blobs=True); x, target = x.to(dev), target.to(dev)
mask = torch.from_numpy(g["masks"][row:row + 1]).to(dev)
P = {k: v.to(dev).clone().requires_grad_(True) for k, v in small_weights().items()}
outs = mmformer_ref.forward(P, x, mask, target, 4.0, "idt", passion, True)
(C.softmax_weighted_loss_bs(outs[0], target).sum() + C.dice_loss_bs(outs[0], target).sum() + outs[1].sum()).backward()
m = _build("mmformer", S).to(dev)
m.is_training, m.mask_type, m.use_passion, m.compute_dtype = True, "idt", passion, torch.float32
m.eval()
o2 = m(x, mask, target=target, temp=4.0)
(criterions.seg_pair_bs(o2[0], target).sum() + o2[1].sum()).backward()
rows = []
for k, p in m.named_parameters():
    a, b = (p.grad.double() if p.grad is not None else torch.zeros_like(p).double()), (P[k].grad.double() if P[k].grad is not None else torch.zeros_like(p).double())
    nb = float(b.norm())
    if nb > 1e-7:
        rows.append((float((a - b).norm()) / nb, k, float(a.norm()), nb))
print(f"row {row} passion={passion}: forward diff {float((o2[0] - outs[0]).abs().max()):.1e}")
pat = sys.argv[3] if len(sys.argv) > 3 else None
if pat:
    rows = [r for r in rows if pat in r[1]]
for e, k, na, nb in (rows if pat else sorted(rows, reverse=True)[:28]):
    print(f"{e:9.2e}  {k:60s} |g| product {na:.6f} oracle {nb:.6f}")
