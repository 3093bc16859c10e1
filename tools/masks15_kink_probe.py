"""LeakyReLU kinks: for every pre-norm conv of one masks15 row (oracle, CPU), the elements whose normalised input x_hat lies within 1e-4 of
zero.  Two f32 implementations differ by ~1e-5 in these activations: such an element can come out on either side of the kink, and its
gradient then differs by the factor 1 / 0.2 -- at the 4^3 / 2^3 levels of the 32^3 fixture one element is 1 / 2048 of a tensor."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, torch.nn.functional as F
from _util import gold, synth, small_weights
from oracle import mmformer_ref
g = gold("masks15_train.npz"); S = int(g["S"])
x, target = synth(1, S, blobs=True)
P = {k: v for k, v in small_weights().items()}
rec = []
orig = mmformer_ref._prenorm_conv
def tap(P_, name, xin, stride=1, pad_mode="reflect"):
    xh = F.instance_norm(xin, eps=1e-5)
    live = xin.flatten(2).std(2, unbiased=False) > 0          # absent modalities: all-zero channels, x_hat = 0 exactly on both sides
    a = xh.abs()[live]
    rec.append((name, tuple(xin.shape), int((a < 1e-4).sum()), int((a < 1e-5).sum()), float(a.min()) if a.numel() else 0.0))
    return orig(P_, name, xin, stride, pad_mode)
mmformer_ref._prenorm_conv = tap
for row in [int(a) for a in sys.argv[1:]] or [1, 3]:
    rec.clear()
    mask = torch.from_numpy(g["masks"][row:row + 1])
    with torch.no_grad():
        mmformer_ref.forward(P, x, mask, target, 4.0, "idt", False, True)
    print(f"row {row} {g['masks'][row].astype(int)}: layers with an |x_hat| < 1e-4 at <= 8^3 voxels:")
    for name, shp, n4, n5, mn in rec:
        if shp[2] <= 8 and n4:
            print(f"   {name:42s} {shp}: {n4} below 1e-4, {n5} below 1e-5, smallest {mn:.1e}")
