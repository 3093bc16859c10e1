"""probe for the mask [0,0,1,0] PASSION step (masks15_train.npz row 1): which loss term carries the 0.4 % of extra gradient"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from _util import gold, grad_norms, synth
from test_masks15_gpu import _build
from passion_amd import train as ptrain
from passion_amd.utils import criterions
g = gold("masks15_train.npz"); S = int(g["S"]); dev = torch.device("cuda:0")
m = _build("mmformer", S).to(dev)
m.is_training, m.mask_type, m.use_passion, m.compute_dtype = True, "idt", True, torch.float32
m.eval()
x, target = synth(1, S, blobs=True); x, target = x.to(dev), target.to(dev)
row = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mask = torch.from_numpy(g["masks"][row:row + 1]).to(dev)
keys = ["multimodal_decode_conv", "t1_pos", "t1_encode_conv", "decoder_fuse", "t1_encoder", "multimodal_transformer"]
ref = dict(zip([str(k) for k in g["mmformer_step_grad_keys"]], g["mmformer_step_grad_norms"][row]))
mk = mask.float()
for what in ("full", "fuse+prm", "kl", "sep", "proto", "fuse", "prm"):
    m.zero_grad(set_to_none=True)
    outs = m(x, mask, target=target, temp=4.0)
    fuse = criterions.seg_pair_bs(outs[0], target).sum()
    terms = {"fuse": fuse, "prm": outs[1].sum(), "kl": 0.5 * (outs[3] * mk).sum(), "sep": (outs[2] * mk).sum(), "proto": 0.1 * (outs[4] * mk).sum()}
    if what == "full":
        loss = ptrain.assemble_loss(outs, target, mask, "idt", True)
    elif what == "fuse+prm":
        loss = terms["fuse"] + terms["prm"]
    else:
        loss = terms[what]
    loss.backward()
    gn = grad_norms((k, p.grad) for k, p in m.named_parameters())
    print(f"{what:9s} loss {float(loss):.6f}", {k: f"{gn.get(k, 0.0):.6f}" for k in keys})
print("reference", {k: f"{ref[k]:.6f}" for k in keys})

# finite differences along the product's own gradient direction of one parameter: the directional derivative must equal the gradient's norm
def total_loss():
    with torch.no_grad():
        outs = m(x, mask, target=target, temp=4.0)
        return float((criterions.seg_pair_bs(outs[0], target).sum() + outs[1].sum()).double())

for pname in ("multimodal_decode_conv.weight", "t1_pos"):
    m.zero_grad(set_to_none=True)
    outs = m(x, mask, target=target, temp=4.0)
    (criterions.seg_pair_bs(outs[0], target).sum() + outs[1].sum()).backward()
    p = dict(m.named_parameters())[pname]
    gvec = p.grad.detach().clone()
    gnorm = float(gvec.double().norm())
    d = gvec / gnorm
    for eps in (2e-2, 1e-2, 5e-3):
        with torch.no_grad():
            p.add_(d, alpha=eps); lp = total_loss()
            p.add_(d, alpha=-2 * eps); lm = total_loss()
            p.add_(d, alpha=eps)
        print(f"{pname}: |g| product {gnorm:.6f}; central difference eps={eps:g}: {(lp - lm) / (2 * eps):.6f}")
