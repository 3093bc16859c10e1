"""is the 'mmformer, t1 alone' deviation the reference's clamp(p, 0.005, 1) meeting interpolation rounding?  Counts, per deep-supervision
head of the oracle run, the up-sampled probabilities ABOVE 1.0 (torch's clamp passes no gradient there) for a few masks15 rows."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, torch.nn.functional as F
from _util import gold, synth, small_weights
from oracle import mmformer_ref
g = gold("masks15_train.npz"); S = int(g["S"])
dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
x, target = synth(1, S, blobs=True); x, target = x.to(dev), target.to(dev)
P = {k: v.to(dev) for k, v in small_weights().items()}
seen = {}
orig = mmformer_ref.decoder_fuse
def tap(P_, xs):
    out = orig(P_, xs); seen["preds"] = out[1]; return out
mmformer_ref.decoder_fuse = tap
for row in [int(a) for a in sys.argv[1:]] or [1, 0, 9, 3]:
    mask = torch.from_numpy(g["masks"][row:row + 1]).to(dev)
    with torch.no_grad():
        mmformer_ref.forward(P, x, mask, target, 4.0, "idt", False, True)
    msg = []
    for lvl, (pred, k) in enumerate(zip(seen["preds"], (2, 4, 8, 16)), 1):
        p = F.interpolate(F.softmax(pred, 1), scale_factor=k, mode="trilinear", align_corners=True)
        tgt = target.float()
        over = (p > 1.0)
        msg.append(f"x{k}: {int(over.sum())} above 1 ({int((over & (tgt > 0)).sum())} on their label), {int((p == 1.0).sum())} equal 1, {int((F.softmax(pred, 1) == 1.0).sum())} saturated before upsampling")
    print(f"row {row} {g['masks'][row].astype(int)}: " + "; ".join(msg))
