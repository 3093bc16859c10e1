"""GPU parity of the RFNet product path: the three RFNet-specific HIP ops against PyTorch CPU float64 references, and
the whole rfnet.Model against the golden vectors of the imported reference (f32 mode, 1e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import gold, grad_norms, rel, synth
from test_rfnet_oracle import TAGS, rfnet_weights

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def nc(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("two", [False, True])
def test_norm_act(dtype, two):
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    xs = [(torch.randn(2, 8, 6, 5, 7, generator=g) * 2 + 1).to(dtype).double() for _ in range(2 if two else 1)]
    xr = [x.clone().requires_grad_(True) for x in xs]
    yr = sum(F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2) for x in xr)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64).to(dtype).double()
    yr.backward(gy)
    xd = [cl(x).to(dev, dtype).requires_grad_(True) for x in xs]
    st = [ops.instnorm_stats(x.detach()) for x in xd]
    y = ops.norm_act(xd[0], st[0], xd[1] if two else None, st[1] if two else None)
    y.backward(cl(gy).to(dev, dtype))
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel(nc(y.detach().float()), yr) < tol
    for a, b in zip(xd, xr):
        assert rel(nc(a.grad.float()), b.grad) < tol * 3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_region_gap_and_mix(dtype):
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    B, C, S = 2, 8, 6
    ys = [torch.randn(B, C, S, S, S, generator=g).to(dtype).double() for _ in range(4)]
    prob = torch.softmax(torch.randn(B, 4, S, S, S, generator=g) * 2, 1).double()
    mask = torch.tensor([[True, False, True, True], [False, True, True, False]])
    w = torch.rand(B, 4, 4, generator=g).double()
    yr = [y.clone().requires_grad_(True) for y in ys]
    wr = w.clone().requires_grad_(True)
    ym = [y * mask[:, m].view(B, 1, 1, 1, 1).double() for m, y in enumerate(yr)]
    Sref = torch.stack([torch.stack([(ym[m] * prob[:, i:i + 1]).sum((2, 3, 4)) for m in range(4)], 1) for i in range(4)], 1)
    Pref = prob.sum((2, 3, 4))
    Rref = torch.cat([prob[:, i:i + 1] * sum(wr[:, i, m].view(B, 1, 1, 1, 1) * ym[m] for m in range(4)) for i in range(4)], 1)
    gS = torch.randn(Sref.shape, generator=g, dtype=torch.float64)
    gR = torch.randn(Rref.shape, generator=g, dtype=torch.float64).to(dtype).double()
    ((Sref * gS).sum() + (Rref * gR).sum()).backward()

    yd = [cl(y).to(dev, dtype).requires_grad_(True) for y in ys]
    wd = w.float().to(dev).requires_grad_(True)
    pd, md = cl(prob).float().to(dev), mask.to(dev)
    Sd, Pd = ops.region_gap(pd, md, yd)
    Rd = ops.region_mix(pd, md, wd, yd)
    ((Sd * gS.float().to(dev)).sum() + (Rd.float() * cl(gR).float().to(dev)).sum()).backward()
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel(Sd, Sref) < tol and rel(Pd, Pref) < 1e-5
    assert rel(nc(Rd.detach().float()), Rref) < tol
    assert rel(wd.grad, wr.grad) < tol * 3
    for a, b in zip(yd, yr):
        assert rel(nc(a.grad.float()), b.grad) < tol * 3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("P", [2, 5, 7])
def test_region_gap_and_mix_masked_passes(dtype, P):
    """P masked passes over the same B samples in one launch (prob / mask / w hold P * B rows, row p * B + b; PASSION's teacher +
    students): outputs per pass, the gradient wrt y is the SUM over the passes (formed inside the backward kernels), dw per pass"""
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    B, C, S = 2, 8, 6
    ys = [torch.randn(B, C, S, S, S, generator=g).to(dtype).double() for _ in range(4)]
    prob = torch.softmax(torch.randn(P * B, 4, S, S, S, generator=g) * 2, 1).double()
    mask = torch.rand(P * B, 4, generator=g) > 0.4
    mask[:, 2] |= ~mask.any(1)
    w = torch.rand(P * B, 4, 4, generator=g).double()
    yr = [y.clone().requires_grad_(True) for y in ys]
    wr = w.clone().requires_grad_(True)
    yv = [y.repeat(P, 1, 1, 1, 1) for y in yr]                                            # virtual sample p * B + b reads y[b]
    ym = [y * mask[:, m].view(P * B, 1, 1, 1, 1).double() for m, y in enumerate(yv)]
    Sref = torch.stack([torch.stack([(ym[m] * prob[:, i:i + 1]).sum((2, 3, 4)) for m in range(4)], 1) for i in range(4)], 1)
    Rref = torch.cat([prob[:, i:i + 1] * sum(wr[:, i, m].view(P * B, 1, 1, 1, 1) * ym[m] for m in range(4)) for i in range(4)], 1)
    gS = torch.randn(Sref.shape, generator=g, dtype=torch.float64)
    gR = torch.randn(Rref.shape, generator=g, dtype=torch.float64).to(dtype).double()
    ((Sref * gS).sum() + (Rref * gR).sum()).backward()

    yd = [cl(y).to(dev, dtype).requires_grad_(True) for y in ys]
    wd = w.float().to(dev).requires_grad_(True)
    pd, md = cl(prob).float().to(dev), mask.to(dev)
    Sd, _ = ops.region_gap(pd, md, yd)
    Rd = ops.region_mix(pd, md, wd, yd)
    ((Sd * gS.float().to(dev)).sum() + (Rd.float() * cl(gR).float().to(dev)).sum()).backward()
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel(Sd, Sref) < tol
    assert rel(nc(Rd.detach().float()), Rref) < tol
    assert rel(wd.grad, wr.grad) < tol * 3
    for a, b in zip(yd, yr):
        assert a.grad.shape == a.shape
        assert rel(nc(a.grad.float()), b.grad) < tol * 3


def build_rfnet():
    from passion_amd.models import rfnet
    torch.manual_seed(1037)
    m = rfnet.Model(num_cls=4)
    m.load_state_dict(rfnet_weights(), strict=True)
    m.eval()
    return m


@pytest.mark.parametrize("tag", TAGS)
def test_rfnet_model_vs_reference_golden(tag):
    from passion_amd import train as ptrain
    dev = _dev()
    g = gold(f"rfnet_{tag}.npz")
    S, B = int(g["S"]), int(g["B"])
    m = build_rfnet().to(dev)
    m.mask_type, m.use_passion, m.is_training = str(g["mask_type"]), bool(g["passion"]), True
    x, target = synth(B, S, blobs=bool(g["blobs"]))
    mask = torch.from_numpy(g["mask"])
    x, target, mask = x.to(dev), target.to(dev), mask.to(dev)
    outs = m(x, mask, target=target, temp=float(g["temp"]))
    loss = ptrain.assemble_loss(outs, target, mask, m.mask_type, m.use_passion)
    pred = outs[0]
    assert rel(pred[:, :, S // 2], g["pred_slice"]) < TOL
    same = torch.from_numpy(pred.argmax(1).cpu().numpy().astype(np.uint8) == g["argmax"])
    assert same.float().mean() > 0.9999
    if not bool(same.all()):
        top2 = pred.detach().cpu().topk(2, dim=1).values
        assert (top2[:, 0] - top2[:, 1])[~same].max() < 1e-5
    for n, o in zip(["prm_loss", "sep_loss", "kl_loss", "proto_loss", "dist"], outs[1:]):
        assert rel(o, g[n]) < TOL, (n, o, g[n])
    assert abs(float(loss) - float(g["loss"])) < TOL * abs(float(g["loss"]))
    loss.backward()
    gn = grad_norms((k, p.grad) for k, p in m.named_parameters())
    for k, ref in zip(g["grad_keys"], g["grad_norms"]):
        assert abs(gn[str(k)] - ref) < 3e-3 * ref + 1e-6, (k, gn[str(k)], ref)
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad::") and np.abs(g[key]).max() > 1e-6:
            assert rel(params[key[6:]].grad, g[key]) < 5e-3, key


def test_rfnet_bf16_tracks():
    from passion_amd import train as ptrain
    dev = _dev()
    g = gold("rfnet_idt_passion_1011.npz")
    m = build_rfnet().to(dev)
    m.mask_type, m.use_passion, m.is_training, m.compute_dtype = "idt", True, True, torch.bfloat16
    x, target = synth(1, 32, blobs=True)
    mask = torch.from_numpy(g["mask"]).to(dev)
    outs = m(x.to(dev), mask, target=target.to(dev), temp=4.0)
    for n, o in zip(["prm_loss", "sep_loss", "kl_loss"], outs[1:4]):
        assert rel(o, g[n]) < 8e-2, n
    ptrain.assemble_loss(outs, target.to(dev), mask, "idt", True).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_rfnet_direct_gradients_change_nothing():
    """train.train_step's direct-gradient mode on RFNet, whose narrow convs run on zero-padded COPIES of their weights
    (blocks.run_conv_auto): a padded copy is not a leaf and must keep sending its gradient through autograd -- two bf16 PASSION
    steps with and without the mode end in bit-identical parameters, and every parameter moved"""
    import copy
    from passion_amd import train as ptrain
    from passion_amd.models import rfnet
    dev = torch.device("cuda:0")
    torch.manual_seed(1037)
    base = rfnet.Model(num_cls=4).to(dev)
    base.mask_type, base.use_passion, base.compute_dtype = "idt", True, torch.bfloat16
    base.train()
    S = 32
    x, target = synth(2, S, blobs=True)
    x, target = x.to(dev), target.to(dev)
    mask = torch.tensor([[True, False, True, True], [True, True, True, False]], device=dev)
    res = []
    keep = ptrain.DIRECT_GRADS
    try:
        for direct in (True, False):
            ptrain.DIRECT_GRADS = direct
            m = copy.deepcopy(base)
            opt = ptrain.make_optimizer(m, lr=1e-3)
            torch.manual_seed(5)
            for _ in range(2):
                ptrain.train_step(m, opt, x, mask, target, 4.0, ptrain.LossState(dev))
            torch.cuda.synchronize()
            res.append(m.state_dict())
    finally:
        ptrain.DIRECT_GRADS = keep
    bad = [k for k in res[0] if not torch.equal(res[0][k], res[1][k])]
    assert not bad, bad[:5]
    sd0 = base.state_dict()
    still = [k for k in res[0] if res[0][k].dtype.is_floating_point and torch.equal(res[0][k], sd0[k])]
    assert not still, still[:5]
