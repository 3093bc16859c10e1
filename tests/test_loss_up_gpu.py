"""The x k upsampling inside the loss kernels (ops.wce_dice_up / ops.kl_probs_up, loss.hip) against (a) the path it replaces --
ops.upsample then the plain loss kernels --, (b) a float64 torch restatement (F.interpolate trilinear align_corners=True + the
criterions formulas, via oracle/criterions_ref.py) with gradients, and (c) the IMPORTED REFERENCE's values in tests/golden/losses.npz
through the public passion_amd.utils.criterions API the reference's train loop calls (incl. up_op)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import gold, rel   # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


@pytest.mark.parametrize("k,S,B,Bt", [(2, 16, 2, 2), (4, 8, 2, 1), (8, 4, 4, 2), (16, 2, 1, 1), (2, 64, 1, 1)])
def test_fused_upsample_losses_match_the_materialised_path_and_float64(k, S, B, Bt):
    from oracle import criterions_ref as C
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(100 + k)
    Sf = S * k
    logits = torch.randn(B, 4, S, S, S, generator=g) * 2.0
    logits_t = torch.randn(Bt, 4, S, S, S, generator=g) * 2.0
    lab = torch.randint(0, 4, (Bt, Sf, Sf, Sf), generator=g)
    target = F.one_hot(lab, 4).permute(0, 4, 1, 2, 3).double()
    tcl = ops.target_to_cl(target.to(dev))
    temp = 4.0
    # ---- float64 reference (teacher / target of student b: b % Bt)
    up = torch.nn.Upsample(scale_factor=k, mode="trilinear", align_corners=True)
    lr = logits.double().requires_grad_(True)
    rep = B // Bt
    t_rep, lt_rep = target.repeat(rep, 1, 1, 1, 1), logits_t.double().repeat(rep, 1, 1, 1, 1)
    pr = torch.softmax(lr, 1)
    wce_r = C.softmax_weighted_loss_bs(pr, t_rep, 4, up)
    dice_r = C.dice_loss_bs(pr, t_rep, 4, up_op=up)
    kl_r = C.temp_kl_loss_bs(lr, lt_rep, t_rep, 4, temp, up)
    gw = torch.tensor([[0.7, -1.3]]).repeat(B, 1).double()
    gk = torch.linspace(0.5, 1.5, B).double().view(B, 1)
    ((torch.cat([wce_r, dice_r], 1) * gw).sum() + (kl_r * gk).sum()).backward()
    # ---- fused and materialised on the GPU
    res = {}
    for fused in (True, False):
        ops.UP_FUSED = fused
        try:
            ld = _cl(logits).to(dev).requires_grad_(True)
            pd = ops.softmax_classes(ld, 1.0, "cl")
            pair = ops.wce_dice_up(pd, tcl, k)
            ps = ops.softmax_classes(ld, 1.0 / temp, "cl")
            pt = ops.softmax_classes(_cl(logits_t).to(dev), 1.0 / temp, "cl")
            kl = ops.kl_probs_up(ps, pt, temp, k)
            ((pair * gw.float().to(dev)).sum() + (kl * gk.float().to(dev)).sum()).backward()
            res[fused] = (pair.detach().cpu(), kl.detach().cpu(), ld.grad.detach().cpu())
        finally:
            ops.UP_FUSED = True
    pair, kl, gl = res[True]
    assert rel(pair[:, 0:1], wce_r) < 1e-5 and rel(pair[:, 1:2], dice_r) < 1e-5 and rel(kl, kl_r) < 1e-5
    assert rel(gl, _cl(lr.grad)) < 2e-4, rel(gl, _cl(lr.grad))
    # the path it replaces: the same numbers up to the f32 order of the sums
    assert rel(pair, res[False][0]) < 2e-6 and rel(kl, res[False][1]) < 2e-6
    assert rel(gl, res[False][2]) < 1e-5, rel(gl, res[False][2])


def test_fused_losses_repeat_bit_for_bit():
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 8, 8, 4, generator=g).to(dev)
    tcl = F.one_hot(torch.randint(0, 4, (2, 32, 32, 32), generator=g), 4).float().to(dev)
    outs = []
    for _ in range(2):
        xd = x.clone().requires_grad_(True)
        p = ops.softmax_classes(xd, 1.0, "cl")
        pair = ops.wce_dice_up(p, tcl, 4)
        kl = ops.kl_probs_up(ops.softmax_classes(xd, 0.25, "cl"), p.detach().flip(0), 4.0, 4)
        (pair.sum() + kl.sum()).backward()
        outs.append((pair.detach().clone(), kl.detach().clone(), xd.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_public_criterions_against_the_reference_values():
    """passion_amd.utils.criterions -- the functions code/train.py:228 and mmformer.py:540-566 call -- on the GPU against the imported
    reference's outputs (tests/golden/losses.npz, tools/make_golden.py), incl. up_op = nn.Upsample x 2 / 4 / 8"""
    from passion_amd.utils import criterions as PC
    dev = _dev()
    g = gold("losses.npz")
    t_ = lambda a: torch.from_numpy(np.asarray(a)).to(dev)      # noqa: E731
    logits, logits_t = t_(g["logits"]), t_(g["logits_t"])
    probs = torch.softmax(logits, 1)
    for tn, key in (("rand", "target"), ("blob", "target_b")):
        t = t_(g[key]).double()
        assert rel(PC.softmax_weighted_loss_bs(probs, t, 4), g[f"wce_{tn}"]) < 1e-5
        assert rel(PC.dice_loss_bs(probs, t, 4), g[f"dice_{tn}"]) < 1e-5
        pl, d = PC.prototype_passion_loss_bs(t_(g["fs"]), t_(g["ft"]), t, None, None, 4, 4.0)
        assert rel(pl, g[f"proto_{tn}"]) < 1e-4 and rel(d, g[f"protodist_{tn}"]) < 1e-4
    t = t_(g["target"]).double()
    assert rel(PC.temp_kl_loss_bs(logits, logits_t, t, 4, 4.0), g["kl_T4"]) < 1e-5
    assert rel(PC.temp_kl_loss_bs(logits, logits_t, t, 4, 1.0), g["kl_T1"]) < 1e-5
    for k in (2, 4, 8):
        up = torch.nn.Upsample(scale_factor=k, mode="trilinear", align_corners=True)
        lo, lo_t = t_(g[f"lo{k}"]), t_(g[f"lo{k}_t"])
        assert rel(PC.softmax_weighted_loss_bs(torch.softmax(lo, 1), t, 4, up), g[f"wce_up{k}"]) < 1e-5
        assert rel(PC.dice_loss_bs(torch.softmax(lo, 1), t, 4, up_op=up), g[f"dice_up{k}"]) < 1e-5
        assert rel(PC.temp_kl_loss_bs(lo, lo_t, t, 4, 4.0, up), g[f"kl_up{k}"]) < 1e-5


_ROWS_SCRIPT = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from passion_amd import ops
dev = torch.device("cuda:0")
out = {}
for k, dims, B, Bt in [(2, (8, 12, 10), 2, 2), (4, (8, 8, 8), 4, 1), (16, (8, 8, 8), 1, 1), (8, (4, 6, 5), 5, 1), (2, (3, 4, 130), 1, 1)]:
    g = torch.Generator().manual_seed(k * 100 + B)
    ps = torch.softmax(torch.randn(B, *dims, 4, generator=g) * 2, -1).to(dev).requires_grad_(True)
    pt = torch.softmax(torch.randn(Bt, *dims, 4, generator=g) * 2, -1).to(dev)
    fine = tuple(d * k for d in dims)
    tcl = torch.nn.functional.one_hot(torch.randint(0, 4, (Bt, *fine), generator=g), 4).float().to(dev).contiguous()
    w = torch.randn(B, 2, generator=g).to(dev)
    (ops.wce_dice_up(ps, tcl, k) * w).sum().backward()
    out[f"wce{k}_{B}"] = ps.grad.clone().cpu()
    ps.grad = None
    (ops.kl_probs_up(ps, pt, 4.0, k) * w[:, 0]).sum().backward()
    out[f"kl{k}_{B}"] = ps.grad.clone().cpu()
torch.save(out, sys.argv[2])
'''


def test_row_parallel_loss_gradient_equals_the_column_walk(tmp_path):
    """up_loss_bwdw_rows_kernel (one thread per fine voxel, whole rows through LDS; the default) against up_loss_bwdw_kernel (a thread walks
    its fine columns; PASSION_UPLOSS_ROWS=0): the coarse-map gradients of both fused losses bit for bit, incl. rows wider than a workgroup"""
    _dev()
    import subprocess
    res = []
    for v in ("0", "1"):
        out = str(tmp_path / f"rows{v}.pt")
        env = dict(os.environ, PASSION_UPLOSS_ROWS=v)
        subprocess.run([sys.executable, "-c", _ROWS_SCRIPT, ROOT, out], check=True, env=env, timeout=600)
        res.append(torch.load(out))
    assert set(res[0]) == set(res[1]) and len(res[0]) == 10
    for k_ in res[0]:
        assert torch.equal(res[0][k_], res[1][k_]), (k_, float((res[0][k_] - res[1][k_]).abs().max()))
        assert float(res[0][k_].abs().max()) > 0
