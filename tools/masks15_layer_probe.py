"""layer-by-layer hunt for the 'mmformer, t1 alone' gradient deviation: the oracle runs one masks15_train.npz row on the GPU
and records input, output and their gradients of every pre-norm conv of decoder_fuse; each recorded layer is then replayed ALONE through
the product's conv op on the same input / upstream gradient and its dx / dW compared."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from _util import gold, synth, small_weights
from oracle import mmformer_ref
from oracle import criterions_ref as C
from passion_amd import ops, _lib as L
g = gold("masks15_train.npz"); S = int(g["S"]); dev = torch.device("cuda:0")
row = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x, target = synth(1, S, blobs=True); x, target = x.to(dev), target.to(dev)
mask = torch.from_numpy(g["masks"][row:row + 1]).to(dev)
P = {k: v.to(dev).clone().requires_grad_(True) for k, v in small_weights().items()}
rec = []
orig = mmformer_ref._prenorm_conv
def tap(P_, name, xin, stride=1, pad_mode="reflect"):
    if name.startswith("decoder_fuse") and xin.requires_grad:
        xin.retain_grad()
    y = orig(P_, name, xin, stride, pad_mode)
    if name.startswith("decoder_fuse"):
        y.retain_grad()
        rec.append((name, xin, y, stride, pad_mode))
    return y
mmformer_ref._prenorm_conv = tap
heads = []
orig_conv = mmformer_ref._conv
def tap_conv(P_, name, xin, stride=1, pad_mode="reflect"):
    y = orig_conv(P_, name, xin, stride, pad_mode)
    if name.startswith("decoder_fuse.seg_"):
        if xin.requires_grad and not xin.is_leaf:
            xin.retain_grad()
        y.retain_grad()
        heads.append((name, xin, y))
    return y
mmformer_ref._conv = tap_conv
outs = mmformer_ref.forward(P, x, mask, target, 4.0, "idt", False, True)
(C.softmax_weighted_loss_bs(outs[0], target).sum() + C.dice_loss_bs(outs[0], target).sum() + outs[1].sum()).backward()
def cl(t): return t.permute(0, 2, 3, 4, 1).contiguous()
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
print(f"row {row}: {len(rec)} pre-norm convs of decoder_fuse recorded")
for name, xin, y, stride, pad_mode in rec:
    if xin.grad is None or y.grad is None:
        continue
    w, b = P[name + ".conv.weight"], P[name + ".conv.bias"]
    k = w.shape[-1]
    B_, Cin, D, H, W = xin.shape
    if Cin % 8:
        continue
    xd = cl(xin.detach()).requires_grad_(True)
    wd, bd = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    st = ops.instnorm_stats(xd.detach())
    yd = ops.conv_block([ops.SegIn(xd, stats=st)], wd, bd, D, H, W, ksize=k, stride=stride,
                        pad_mode=L.PAD_REFLECT if (pad_mode == "reflect" and k == 3) else L.PAD_ZEROS)
    yd.backward(cl(y.grad))
    dx_ref = cl(xin.grad)
    xs = xin.detach()
    sd = xs.std((2, 3, 4), unbiased=False)[0]
    ratio = float((xs.mean((2, 3, 4))[0].abs() / (sd + 1e-20)).max())
    print(f"{name:38s} {tuple(xin.shape)} k{k} fwd {rel(yd.permute(0, 4, 1, 2, 3), y):.1e} dx {rel(xd.grad, dx_ref):.1e} dW {rel(wd.grad, w.grad):.1e}"
          f"  min std {float(sd.min()):.2e} max |mean|/std {ratio:.1f}")

# the logit heads: plain 1x1x1 convs (no norm in front), f32 logits.  Their input gradient adds to the feature map's other consumers in
# autograd, so the oracle's xin.grad is the TOTAL: compare against the head's own share = conv_transpose of its dY
for name, xin, y in heads:
    w, b = P[name + ".weight"], P[name + ".bias"]
    B_, Cin, D, H, W = xin.shape
    xd = cl(xin.detach()).requires_grad_(True)
    wd, bd = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yd = ops.conv_block([ops.SegIn(xd)], wd, bd, D, H, W, ksize=1, stride=1, pad_mode=L.PAD_ZEROS, out_dtype=torch.float32)
    yd.backward(cl(y.grad))
    dx_own = torch.einsum("bodhw,oi->bidhw", y.grad.double(), w.detach().double()[:, :, 0, 0, 0])
    dw_own = torch.einsum("bodhw,bidhw->oi", y.grad.double(), xin.detach().double())
    print(f"{name:38s} {tuple(xin.shape)} fwd {rel(yd.permute(0, 4, 1, 2, 3), y):.1e} dx {rel(xd.grad.permute(0, 4, 1, 2, 3), dx_own):.1e} dW {rel(wd.grad[:, :, 0, 0, 0], dw_own):.1e}")
