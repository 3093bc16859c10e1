"""per-subset relative gradient-norm errors of the product against masks15_train.npz (debug aid for tests/test_masks15_gpu.py)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from _util import gold, grad_norms, synth
from test_masks15_gpu import _build
from passion_amd import train as ptrain
name = sys.argv[1] if len(sys.argv) > 1 else "mmformer"
dtype = torch.float32
g = gold("masks15_train.npz"); S = int(g["S"]); dev = torch.device("cuda:0")
m = _build(name, S).to(dev)
m.is_training, m.mask_type, m.use_passion, m.compute_dtype = True, "idt", True, dtype
m.eval()
x, target = synth(1, S, blobs=True); x, target = x.to(dev), target.to(dev)
for row in range(15):
    mask = torch.from_numpy(g["masks"][row:row + 1]).to(dev)
    m.zero_grad(set_to_none=True)
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", True)
    loss.backward()
    gn = grad_norms((k, p.grad) for k, p in m.named_parameters())
    errs = {str(k): abs(gn.get(str(k), 0.0) - r) / r for k, r in zip(g[name + "_step_grad_keys"], g[name + "_step_grad_norms"][row]) if r > 0}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print(row, g["masks"][row].astype(int), f"loss err {abs(float(loss) - g[name + '_step_loss'][row]) / g[name + '_step_loss'][row]:.1e}", [(k, f"{v:.1e}") for k, v in top])
    if name == "mmformer" and row == 1:
        p = dict(m.named_parameters())
        print("   decode conv: |dW|", float(p["multimodal_decode_conv.weight"].grad.double().norm()), "|db|", float(p["multimodal_decode_conv.bias"].grad.double().norm()))
