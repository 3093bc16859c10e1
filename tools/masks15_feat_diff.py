"""forward activations of decoder_fuse (five feature maps, four deep-supervision logits), product vs oracle on the GPU, one masks15 row"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from _util import gold, synth, small_weights
from test_masks15_gpu import _build
from oracle import mmformer_ref
g = gold("masks15_train.npz"); S = int(g["S"]); dev = torch.device("cuda:0")
x, target = synth(1, S, blobs=True); x, target = x.to(dev), target.to(dev)
P = {k: v.to(dev) for k, v in small_weights().items()}
seen, seen_p = {}, {}
orig = mmformer_ref.decoder_fuse
def tap(P_, xs):
    out = orig(P_, xs); seen["o"] = out; seen["xs"] = xs; return out
mmformer_ref.decoder_fuse = tap
m = _build("mmformer", S).to(dev)
m.is_training, m.mask_type, m.use_passion, m.compute_dtype = True, "idt", False, torch.float32
m.eval()
pf = m.decoder_fuse.forward
def ptap(*a, **k):
    out = pf(*a, **k); seen_p["o"] = out; seen_p["a"] = a; return out
m.decoder_fuse.forward = ptap
def t_of(f):
    t = f if torch.is_tensor(f) else f.t
    return t.permute(0, 4, 1, 2, 3) if t.dim() == 5 and t.shape[-1] != t.shape[-2] else t
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
for row in [int(a) for a in sys.argv[1:]] or [1, 3]:
    mask = torch.from_numpy(g["masks"][row:row + 1]).to(dev)
    with torch.no_grad():
        mmformer_ref.forward(P, x, mask, target, 4.0, "idt", False, True)
        m(x, mask, target=target, temp=4.0)
    lo, po = seen["o"], seen_p["o"]
    print(f"row {row}: logits {rel(t_of(po[0]), lo[0]):.1e}; deep-supervision logits " + " ".join(f"{rel(t_of(a), b):.1e}" for a, b in zip(po[1], lo[1]))
          + "; features (levels 1..5) " + " ".join(f"{rel(t_of(a), b):.1e}" for a, b in zip(po[2], lo[2])))
    xs_p = seen_p["a"]
    print("   inputs of decoder_fuse: x5 " + f"{rel(t_of(xs_p[4]), seen['xs'][4]):.1e}")
