"""GPU parity of every HIP op (through the C ABI) against plain PyTorch CPU float64 references of
the reference's ops (F.conv3d / F.instance_norm / F.leaky_relu / F.interpolate / softmax / the
criterions formulas).  Tolerances: f32 storage 2e-4 relative to the tensor scale, bf16 storage 3e-2."""
import itertools

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def cl(x):  # NCDHW -> channels-last 5-D
    return x.permute(0, 2, 3, 4, 1).contiguous()


def nc(x):  # channels-last -> NCDHW
    return x.permute(0, 4, 1, 2, 3).contiguous()


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}


def ref_conv_block(xs, norms, masks, weight, bias, k, stride, pad_mode, residual):
    parts = []
    for x, nrm, m in zip(xs, norms, masks):
        xm = x if m is None else x * m.view(-1, 1, 1, 1, 1).to(x.dtype)
        parts.append(F.leaky_relu(F.instance_norm(xm, eps=1e-5), 0.2) if nrm else xm)
    inp = torch.cat(parts, 1)
    if k == 3:
        inp = F.pad(inp, (1,) * 6, mode="reflect" if pad_mode == 1 else "constant")
    y = F.conv3d(inp, weight, bias, stride=stride)
    if residual is not None:
        y = y + residual
    return y


CONV_CASES = [
    # (segC list, Cout, k, stride, pad_mode, (D,H,W), norm, use_mask, residual)
    ([8], 8, 3, 1, 1, (8, 8, 8), True, False, True),
    ([8], 16, 3, 2, 1, (8, 16, 8), True, False, False),
    ([8, 8], 8, 3, 1, 1, (4, 8, 16), True, False, False),
    ([8], 8, 3, 1, 0, (6, 10, 12), True, False, False),
    ([8, 8, 8, 8], 8, 1, 1, 0, (8, 8, 8), True, True, False),
    ([16], 4, 1, 1, 0, (5, 7, 9), False, False, False),
    ([32], 32, 3, 1, 1, (4, 4, 4), True, False, False),
    ([128], 128, 3, 1, 1, (2, 2, 2), True, False, False),
    ([64], 128, 3, 2, 1, (4, 4, 4), True, False, False),
    ([128, 128], 64, 3, 1, 1, (4, 4, 4), True, False, False),
    ([2048], 512, 1, 1, 0, (2, 2, 2), False, False, False),
    ([8], 8, 3, 1, 1, (3, 5, 2), True, False, False),
    # 33 .. 64 blocks of the K axis (the level-5 fusion block's 4 x 128 -> 128 pointwise conv; lean kernels, 32-channel chunks)
    ([128, 128, 128, 128], 128, 1, 1, 0, (8, 8, 8), True, True, False),
    ([128, 128, 128, 128], 64, 3, 1, 1, (4, 4, 8), True, True, False),
    ([256, 128], 32, 3, 1, 0, (4, 8, 8), True, False, False),
    # levels 3..5 at their sizes in a 128^3 step and in the reference's 80^3 crop (conv_deep.hip: K chunks side by side)
    ([32], 32, 3, 1, 1, (32, 32, 32), True, False, True),
    ([64], 64, 3, 1, 1, (16, 16, 16), True, False, False),
    ([128], 128, 3, 1, 1, (8, 8, 8), True, False, True),
    ([64, 64], 64, 3, 1, 1, (16, 16, 16), True, True, False),
    ([128], 128, 3, 1, 0, (8, 8, 8), True, False, False),
    ([128], 128, 3, 1, 1, (5, 5, 5), True, False, False),
    ([64], 64, 3, 1, 1, (10, 10, 10), True, False, False),
    ([32, 32], 32, 3, 1, 1, (20, 20, 20), True, False, False),
    ([64], 64, 1, 1, 0, (16, 16, 16), True, False, False),
    ([64, 64, 64, 64], 64, 1, 1, 0, (10, 10, 10), True, True, False),
    ([128], 128, 3, 1, 1, (10, 15, 15), True, False, True),          # levels 5 / 4 of a 240 x 240 x 160 volume (depth-sharded slabs: few planes)
    ([64, 64], 64, 3, 1, 1, (6, 30, 30), True, True, False),
    ([64], 32, 3, 1, 0, (7, 30, 30), True, False, False),
    # the finest level's 3x3x3 layers (8 output channels, H % 8 = W % 16 = 0, D >= 8): finish-free backward -- weight gradient with
    # the indicator slab first, means from it, fold-free marching input gradient (conv_march.hip FIN); one / two tiles per axis
    ([8], 8, 3, 1, 1, (8, 8, 16), True, False, False),
    ([16], 8, 3, 1, 1, (16, 16, 32), True, False, True),
    ([8, 8], 8, 3, 1, 1, (8, 16, 16), True, True, False),
    ([8], 8, 3, 1, 1, (24, 24, 48), True, False, False),
    # 8-channel K axis, <= 8 rows: the depth-pair kernel (conv_pair.hip) -- 32-wide forward tiles with a residual, a masked
    # modality, zero padding (no halo outputs), an odd depth (falls back to conv_march.hip), four output channels, no statistics
    ([8], 8, 3, 1, 1, (16, 16, 32), True, False, True),
    ([8], 8, 3, 1, 1, (10, 8, 32), True, True, False),
    ([8], 8, 3, 1, 0, (8, 16, 32), True, False, False),
    ([8], 8, 3, 1, 1, (7, 8, 16), True, False, False),
    ([8], 4, 3, 1, 1, (8, 8, 16), True, False, False),
    ([8], 8, 3, 1, 1, (12, 16, 16), False, False, True),
    ([8], 8, 3, 1, 1, (40, 32, 32), True, False, False),
    # the encoders' first stride-2 layer (8 -> 16): marching gather-form input gradient (conv_s2_dgrad.hip, conv_s2_march_kernel)
    ([8], 16, 3, 2, 1, (16, 32, 48), True, False, False),
    ([8], 16, 3, 2, 1, (36, 20, 34), True, True, False),
    ([8], 16, 3, 2, 1, (6, 8, 8), False, False, False),
    # sixteen output channels on marching-eligible volumes: wgrad_march16_kernel (conv_wgrad_march.hip)
    ([16], 16, 3, 1, 1, (16, 16, 32), True, False, True),
    ([16, 16], 16, 3, 1, 1, (8, 16, 16), True, True, False),
    ([16], 16, 3, 1, 0, (9, 8, 16), True, False, False),
    ([8, 8, 8], 16, 3, 1, 1, (12, 24, 16), False, False, False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(i) for i in range(len(CONV_CASES))])
def test_conv_block_fwd_bwd(case, dtype):
    from passion_amd import ops
    dev = _dev()
    segC, Cout, k, stride, pad_mode, (D, H, W), norm, use_mask, use_res = case
    B = 2
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(B, c, D, H, W, generator=g, dtype=torch.float64) * 1.5 + 0.3 for c in segC]
    cin = sum(segC)
    w = torch.randn(Cout, cin, k, k, k, generator=g, dtype=torch.float64) / (cin * k ** 3) ** 0.5
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    mask = None
    if use_mask:
        mask = torch.tensor([[True, False, True, True], [False, True, True, False]])[:, :len(segC)]
    # quantise inputs to the storage dtype so both sides see identical values
    xs = [x.to(dtype).double() for x in xs]
    pad = k // 2
    Do, Ho, Wo = [(s + 2 * pad - k) // stride + 1 for s in (D, H, W)]
    res = (torch.randn(B, Cout, Do, Ho, Wo, generator=g, dtype=torch.float64).to(dtype).double()
           if use_res else None)
    gy = torch.randn(B, Cout, Do, Ho, Wo, generator=g, dtype=torch.float64).to(dtype).double()

    xr = [x.clone().requires_grad_(True) for x in xs]
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    masks = [None if mask is None else mask[:, i].double() for i in range(len(segC))]
    yr = ref_conv_block(xr, [norm] * len(segC), masks, wr, br, k, stride, pad_mode, res)
    yr.backward(gy)

    xd = [cl(x).to(dev, dtype).requires_grad_(True) for x in xs]
    wd = w.float().to(dev).requires_grad_(True)
    bd = b.float().to(dev).requires_grad_(True)
    md = mask.to(dev) if mask is not None else None
    segs = [ops.SegIn(x, stats=ops.instnorm_stats(x.detach()) if norm else None, mask=md, mask_col=i)
            for i, x in enumerate(xd)]
    resd = cl(res).to(dev, dtype) if res is not None else None
    y = ops.conv_block(segs, wd, bd, D, H, W, ksize=k, stride=stride, pad_mode=pad_mode, residual=resd)
    y.backward(cl(gy).to(dev, dtype))
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert relerr(nc(y.detach().float()), yr) < tol, f"fwd {relerr(nc(y.detach().float()), yr)}"
    assert relerr(wd.grad, wr.grad) < tol * 2, f"dW {relerr(wd.grad, wr.grad)}"
    assert relerr(bd.grad, br.grad) < tol * 2, f"db {relerr(bd.grad, br.grad)}"
    for i in range(len(segC)):
        e = relerr(nc(xd[i].grad.float()), xr[i].grad)
        assert e < tol * 3, f"dx[{i}] {e}"


@pytest.mark.parametrize("shape", [([8], (8, 8, 16)), ([8, 8], (16, 16, 32)), ([16], (40, 24, 48)), ([8], (130, 16, 32)), ([8], (24, 24, 48)),
                                   ([8], (16, 16, 32)), ([8], (10, 32, 64))])
@pytest.mark.parametrize("norm", [True, False], ids=["prenorm", "raw_input"])
def test_finish_free_marching_backward_equals_dgrad_plus_finish(shape, norm):
    """the finest level's backward in its two forms on the same inputs: (weight gradient with indicator slab -> means -> fold-free
    marching input gradient with the final epilogue) against (weight gradient, padded-domain input gradient, finish kernel): input
    gradients to the bf16 rounding of the stored values, weight / bias gradients to the f32 summation order; twice for bit equality.
    raw_input: nothing normalised in the loader (RFNet's post-norm convs) -- the final-form input gradient alone replaces the
    padded-domain gradient + the finish pass that only folded its halo"""
    from passion_amd import ops
    dev = _dev()
    segC, (D, H, W) = shape
    dtype, B = torch.bfloat16, 3
    g = torch.Generator().manual_seed(41 + D)
    xs = [(torch.randn(B, D, H, W, c, generator=g) * 1.5 + 0.3).to(dtype) for c in segC]
    cin = sum(segC)
    w = torch.randn(8, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5
    b = torch.randn(8, generator=g)
    gy = torch.randn(B, D, H, W, 8, generator=g).to(dtype)
    mask = torch.tensor([[True, True], [False, True], [True, False]])[:, :len(segC)].to(dev) if len(segC) > 1 else None
    res = {}
    for mode in (1, 0, 1):
        old = ops.set_option("conv_marchfinal", mode)
        try:
            xd = [x.to(dev).requires_grad_(True) for x in xs]
            wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
            segs = [ops.SegIn(x, stats=ops.instnorm_stats(x.detach()) if norm else None, mask=mask, mask_col=i) for i, x in enumerate(xd)]
            y = ops.conv_block(segs, wd, bd, D, H, W, ksize=3, stride=1, pad_mode=1)
            y.backward(gy.to(dev))
            torch.cuda.synchronize()
            out = (wd.grad.clone(), bd.grad.clone(), [x.grad.clone() for x in xd])
            if mode in res:
                assert torch.equal(out[0], res[mode][0]) and all(torch.equal(a, c) for a, c in zip(out[2], res[mode][2]))
            res[mode] = out
        finally:
            ops.set_option("conv_marchfinal", old)
    assert relerr(res[1][0], res[0][0]) < 1e-4 and relerr(res[1][1], res[0][1]) < 1e-4
    for a, c in zip(res[1][2], res[0][2]):
        assert relerr(a.float(), c.float()) < 1.5e-2, relerr(a.float(), c.float())


PW_CASES = [
    # (segC list, Cout, (D,H,W), norm, use_mask): 1x1x1 convs over >= 32768 voxels -> conv_pw.hip (LDS-free streaming GEMM)
    ([8, 8, 8, 8], 8, (32, 32, 32), True, True),
    ([16], 16, (33, 32, 40), True, False),
    ([16, 16, 16, 16], 16, (32, 32, 40), True, True),
    ([32, 32, 32], 32, (32, 32, 32), True, False),
    ([8], 8, (64, 64, 64), False, False),
    ([128], 32, (32, 32, 33), True, False),
    ([8], 8, (33, 40, 32), True, True),               # 8-channel K axis: the 32x32x8 MFMA form (forward and gradient)
    ([16], 8, (32, 32, 32), True, False),             # its gradient has 16 rows over a K of 8
    ([8, 8, 8], 8, (32, 32, 32), True, True),         # gradient: 24 rows (three segments) over a K of 8
]


@pytest.mark.parametrize("case", PW_CASES, ids=[str(i) for i in range(len(PW_CASES))])
def test_pointwise_streaming_kernel(case):
    """conv_pw.hip against the float64 reference (forward; its backward stays on the lean kernels and is checked through the
    same call), and against the lean kernel it replaces: same bf16 inputs, same MFMA products, so the two agree to the rounding of
    the f32 accumulation order"""
    from passion_amd import ops
    dev = _dev()
    segC, Cout, (D, H, W), norm, use_mask = case
    dtype, B = torch.bfloat16, 2
    g = torch.Generator().manual_seed(23 + Cout)
    xs = [(torch.randn(B, c, D, H, W, generator=g, dtype=torch.float64) * 1.5 + 0.3).to(dtype).double() for c in segC]
    cin = sum(segC)
    w = torch.randn(Cout, cin, 1, 1, 1, generator=g, dtype=torch.float64) / cin ** 0.5
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    mask = torch.tensor([[True, False, True, True], [False, True, True, False]])[:, :len(segC)] if use_mask else None
    gy = torch.randn(B, Cout, D, H, W, generator=g, dtype=torch.float64).to(dtype).double()
    xr = [x.clone().requires_grad_(True) for x in xs]
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    masks = [None if mask is None else mask[:, i].double() for i in range(len(segC))]
    yr = ref_conv_block(xr, [norm] * len(segC), masks, wr, br, 1, 1, 0, None)
    yr.backward(gy)
    md = mask.to(dev) if mask is not None else None
    res = {}
    for pw in (1, 0, 2):                       # 2: streaming kernels, but dgrad + finish instead of the finish-free backward
        old = ops.set_option("conv_pw", 1 if pw else 0)
        oldf = ops.set_option("conv_pwfinal", 0 if pw == 2 else 1)
        try:
            xd = [cl(x).to(dev, dtype).requires_grad_(True) for x in xs]
            wd, bd = w.float().to(dev).requires_grad_(True), b.float().to(dev).requires_grad_(True)
            segs = [ops.SegIn(x, stats=ops.instnorm_stats(x.detach()) if norm else None, mask=md, mask_col=i) for i, x in enumerate(xd)]
            y, st = ops.conv_block(segs, wd, bd, D, H, W, ksize=1, stride=1, pad_mode=0, want_stats=True)
            y.backward(cl(gy).to(dev, dtype))
            res[pw] = (y.detach().clone(), st.clone(), wd.grad.clone(), [x.grad.clone() for x in xd], bd.grad.clone())
        finally:
            ops.set_option("conv_pw", old)
            ops.set_option("conv_pwfinal", oldf)
    y, st, dw, dxs, db = res[1]
    assert relerr(nc(y.float()), yr) < TOL[dtype], relerr(nc(y.float()), yr)
    assert relerr(dw, wr.grad) < 2 * TOL[dtype] and relerr(db, br.grad) < 2 * TOL[dtype]
    # the finish-free backward (InstanceNorm-backward means from the weight gradient: conv_pw.hip) against dgrad + finish: the same
    # input gradient up to the bf16 rounding of the stored values, the same weight / bias gradient up to the f32 summation order
    for a, c in zip(dxs, res[2][3]):
        assert relerr(a.float(), c.float()) < 1e-2, relerr(a.float(), c.float())
    assert relerr(dw, res[2][2]) < 1e-4 and relerr(db, res[2][4]) < 1e-4
    for i in range(len(segC)):
        assert relerr(nc(dxs[i].float()), xr[i].grad) < 3 * TOL[dtype]
    # against the lean kernel: identical up to the order of the f32 accumulation (one bf16 rounding step at most, rarely)
    d = (y.float() - res[0][0].float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(y.float().abs().max()) and float((d > 0).float().mean()) < 0.05
    want = ops.instnorm_stats(y)
    std = 1.0 / want[..., 1]
    assert float(((st[..., 0] - want[..., 0]).abs() / (std + want[..., 0].abs())).max()) < 2e-5
    assert float(((st[..., 1] - want[..., 1]).abs() / want[..., 1]).max()) < 1e-4


WPW_CASES = PW_CASES + [
    # ragged volumes (voxel counts that are no multiple of the 64-voxel chunks / of the slots), one and two samples' worth of slots
    ([8], 8, (5, 7, 9), True, True),
    ([16, 16], 32, (3, 11, 13), True, False),
    ([16, 16, 16, 16], 24, (9, 16, 17), True, True),
    ([32], 16, (4, 4, 4), False, False),
    # >= 8 M input elements: conv.hip plans the finish-free backward, the weight gradient also leaves the indicator sums (IDENT)
    ([8], 8, (64, 64, 128), True, False),
    ([8, 8, 8, 8], 8, (32, 64, 64), True, True),
    ([16], 16, (64, 64, 64), True, False),
]


@pytest.mark.parametrize("f32_head", [False, True], ids=["bf16_dy", "f32_logits"])
@pytest.mark.parametrize("case", WPW_CASES, ids=[str(i) for i in range(len(WPW_CASES))])
def test_streaming_pointwise_weight_gradient_equals_the_lean_kernel(case, f32_head):
    """conv_wgrad_pw.hip (a wavefront streams 64 voxels at a time, every input block of a voxel in one workgroup) against
    wgrad_lean_kernel<1, *>: the same bf16 operands and MFMA products, another order of the f32 sums -- weight, bias and (through the
    indicator sums of the finish-free backward) input gradients; and against the float64 reference"""
    from passion_amd import ops
    dev = _dev()
    segC, Cout, (D, H, W), norm, use_mask = case
    if f32_head:
        if Cout > 8:
            pytest.skip("f32 logits: <= 8 classes")
        Cout = 4
    dtype, B = torch.bfloat16, 2
    g = torch.Generator().manual_seed(101 + Cout + len(segC))
    xs = [(torch.randn(B, c, D, H, W, generator=g, dtype=torch.float64) * 1.5 + 0.3).to(dtype).double() for c in segC]
    cin = sum(segC)
    w = torch.randn(Cout, cin, 1, 1, 1, generator=g, dtype=torch.float64) / cin ** 0.5
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    cols = torch.tensor([[True, False, True, True, False, True, True, True], [False, True, True, False, True, True, False, True]])
    mask = cols[:, :len(segC)] if use_mask else None
    # a gradient with a mean: the InstanceNorm-backward means (which the finish-free backward takes from this kernel's indicator
    # sums) are then O(1) terms of dx, not 1 / sqrt(V) ones -- a wrong indicator column shows
    gy = torch.randn(B, Cout, D, H, W, generator=g, dtype=torch.float64) * 0.5 + torch.linspace(0.5, 1.5, Cout, dtype=torch.float64).view(1, -1, 1, 1, 1)
    gy = gy.float().double() if f32_head else gy.to(dtype).double()
    xr = [x.clone().requires_grad_(True) for x in xs]
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    masks = [None if mask is None else mask[:, i].double() for i in range(len(segC))]
    ref_conv_block(xr, [norm] * len(segC), masks, wr, br, 1, 1, 0, None).backward(gy)
    md = mask.to(dev) if mask is not None else None
    res = {}
    oldmin = ops.set_option("wgrad_pw_minvox", 0)      # the product leaves launches under 2^20 voxels on the lean kernel
    for on in (1, 0):
        old = ops.set_option("wgrad_pw", on)
        try:
            xd = [cl(x).to(dev, dtype).requires_grad_(True) for x in xs]
            wd, bd = w.float().to(dev).requires_grad_(True), b.float().to(dev).requires_grad_(True)
            segs = [ops.SegIn(x, stats=ops.instnorm_stats(x.detach()) if norm else None, mask=md, mask_col=i) for i, x in enumerate(xd)]
            y = ops.conv_block(segs, wd, bd, D, H, W, ksize=1, stride=1, pad_mode=0, out_dtype=torch.float32 if f32_head else None)
            y.backward(cl(gy).to(dev, torch.float32 if f32_head else dtype))
            res[on] = (wd.grad.clone(), bd.grad.clone(), [x.grad.clone() for x in xd])
        finally:
            ops.set_option("wgrad_pw", old)
    ops.set_option("wgrad_pw_minvox", oldmin)
    dw, db, dxs = res[1]
    assert relerr(dw, wr.grad) < 2 * TOL[dtype] and relerr(db, br.grad) < 2 * TOL[dtype]
    assert relerr(dw, res[0][0]) < 1e-4 and relerr(db, res[0][1]) < 1e-4, (relerr(dw, res[0][0]), relerr(db, res[0][1]))
    for a, c, r in zip(dxs, res[0][2], xr):
        assert relerr(a.float(), c.float()) < 5e-3, relerr(a.float(), c.float())      # one bf16 step of the stored gradient at most
        assert relerr(nc(a.float()), r.grad) < 3 * TOL[dtype]


S2_CASES = [
    # (Cin, Cout, (D,H,W), normalised input, modality mask): the stride-2 convs of encoder levels 2..5 (mmformer.py:33,38,43,48) and
    # ragged / non-cube volumes; conv_s2_dgrad.hip (gather form: 1-8 taps per parity class) vs the zero-stuffed lean kernel
    (8, 16, (16, 16, 16), True, False),
    (8, 16, (10, 12, 20), True, True),
    (16, 32, (8, 24, 16), True, False),
    (32, 64, (8, 8, 8), True, False),
    (64, 128, (4, 4, 4), True, False),
    (8, 16, (6, 18, 30), False, False),
    (16, 32, (32, 32, 32), True, False),
]


@pytest.mark.parametrize("case", S2_CASES, ids=[str(i) for i in range(len(S2_CASES))])
def test_stride2_dgrad_gather_kernel(case):
    from passion_amd import ops
    dev = _dev()
    Cin, Cout, (D, H, W), norm, use_mask = case
    dtype = torch.bfloat16
    B = 3
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(B, Cin, D, H, W, generator=g, dtype=torch.float64) * 1.5 + 0.3).to(dtype).double()
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g, dtype=torch.float64) / (Cin * 27) ** 0.5
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    mask = torch.tensor([[True], [False], [True]]) if use_mask else None
    gy = torch.randn(B, Cout, D // 2, H // 2, W // 2, generator=g, dtype=torch.float64).to(dtype).double()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = ref_conv_block([xr], [norm], [None if mask is None else mask[:, 0].double()], wr, br, 3, 2, 1, None)
    yr.backward(gy)
    got = {}
    for s2 in (1, 0):
        old = ops.set_option("conv_s2", s2)
        try:
            xd = cl(x).to(dev, dtype).requires_grad_(True)
            wd, bd = w.float().to(dev).requires_grad_(True), b.float().to(dev).requires_grad_(True)
            md = mask.to(dev) if mask is not None else None
            seg = ops.SegIn(xd, stats=ops.instnorm_stats(xd.detach()) if norm else None, mask=md, mask_col=0)
            y = ops.conv_block([seg], wd, bd, D, H, W, ksize=3, stride=2, pad_mode=1)
            y.backward(cl(gy).to(dev, dtype))
            torch.cuda.synchronize()
            got[s2] = nc(xd.grad.float()).cpu()
        finally:
            ops.set_option("conv_s2", old)
    tol = TOL[dtype]
    e1, e0 = relerr(got[1], xr.grad), relerr(got[0], xr.grad)
    assert e1 < tol * 3, f"gather-form dx {e1} (zero-stuffed kernel: {e0})"
    assert e1 < 1.5 * e0 + 2e-3                      # same bf16 operands, same f32 accumulation: as close as the kernel it replaces
    if use_mask:
        assert float(got[1][1].abs().max()) == 0.0   # absent modality: no gradient


PATH_CASES = [
    # bigger volumes: several bricks per persistent workgroup, interior + border + partial bricks
    ([8], 8, 3, 1, 1, (12, 24, 40), True, False, True),
    ([8, 8], 16, 3, 1, 1, (9, 20, 28), True, True, False),
    ([32], 32, 3, 1, 1, (8, 16, 24), True, False, False),
    ([8, 8, 8, 8], 8, 1, 1, 0, (12, 16, 24), True, True, False),
    ([16], 16, 3, 1, 0, (8, 24, 24), True, False, False),
    ([8], 8, 3, 1, 1, (12, 16, 24), False, False, False),      # reflect pad, input NOT normalised (fold-only finish)
]
ZERO_PAD_MARCH_CASES = [
    # zero-padded 3x3x3 convs on volumes the marching kernel takes: padded-domain dgrad without halo outputs, finish without fold
    ([8], 8, 3, 1, 0, (12, 16, 32), True, False, False),
    ([8, 8], 16, 3, 1, 0, (9, 24, 16), True, True, True),
    ([16], 8, 3, 1, 0, (6, 8, 48), True, False, False),
]


@pytest.mark.parametrize("case", ZERO_PAD_MARCH_CASES, ids=[str(i) for i in range(len(ZERO_PAD_MARCH_CASES))])
def test_zero_padded_dgrad_on_the_marching_kernel(case):
    from passion_amd import ops
    ops.prof_read()
    ops.prof_enable(1 << 2)
    try:
        test_conv_block_fwd_bwd(case, torch.bfloat16)
    finally:
        ops.prof_enable(0)
    names = {r[4] for r in ops.prof_read()}
    assert any(n.startswith("conv_march_kernel") or n.startswith("conv_pair_kernel") for n in names), names      # (8 rows: the depth-pair form)


@pytest.mark.parametrize("case", [([16, 16], 16, 3, 1, 1, (32, 64, 128), True, True, True), ([32], 8, 3, 1, 0, (64, 64, 64), True, False, False)],
                         ids=["cat16+16", "32to8_zero_pad"])
def test_marching_forward_with_32_input_channels(case):
    """from 2^19 voxels up a 32-channel K axis runs the marching forward with four 8-channel blocks (conv_march_kernel<4, 16, 0, *>:
    the decoders' 32 -> 16 layers at 64^3, RFNet's 32-channel layers at 128^3) -- against the f64 reference like every other path"""
    from passion_amd import ops
    ops.prof_read()
    ops.prof_enable(1 << 0)
    try:
        test_conv_block_fwd_bwd(case, torch.bfloat16)
    finally:
        ops.prof_enable(0)
    names = {r[4] for r in ops.prof_read()}
    assert any(n.startswith("conv_march_kernel<4, 16, 0") for n in names), names


PATHS = {"lean": {}, "lean_mirror_taps": {"conv_padout": 0}, "general": {"conv_lean": 0}, "few_workgroups": {"conv_target": 8}}


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("case", PATH_CASES, ids=[str(i) for i in range(len(PATH_CASES))])
def test_conv_kernel_paths(case, path):
    """the same bf16 conv through the lean persistent kernels (default), the lean dgrad with mirror taps instead of the
    padded domain, the general one-brick-per-workgroup kernels, and with so few persistent workgroups that each walks
    many bricks -- all against the f64 reference"""
    from passion_amd import ops
    dev = _dev()
    old = {k: ops.set_option(k, v) for k, v in PATHS[path].items()}
    try:
        test_conv_block_fwd_bwd(case, torch.bfloat16)
    finally:
        for k, v in old.items():
            ops.set_option(k, v)


@pytest.mark.parametrize("segC,Cout,k,norm,dims", [([16], 4, 1, False, (8, 16, 24)), ([8, 8], 4, 1, True, (8, 16, 24)), ([8], 4, 3, True, (8, 16, 24)),
                                                   ([32], 8, 1, False, (8, 16, 24)),
                                                   # >= 32768 voxels: the streaming pointwise kernels (conv_pw.hip; 8 channels: 32x32x8 MFMA form)
                                                   ([8], 4, 1, True, (32, 32, 33)), ([16], 4, 1, True, (33, 32, 32)), ([8], 8, 1, False, (32, 32, 32)),
                                                   ([8, 8, 8, 8], 4, 1, True, (32, 32, 32))])
def test_conv_f32_logit_head(segC, Cout, k, norm, dims):
    """bf16 activations -> f32 logits (segmentation / region-probability heads): forward, dgrad with f32 dY, wgrad"""
    from passion_amd import ops
    dev = _dev()
    B, (D, H, W) = 2, dims
    g = torch.Generator().manual_seed(13)
    xs = [(torch.randn(B, c, D, H, W, generator=g, dtype=torch.float64) * 1.2 + 0.2).to(torch.bfloat16).double() for c in segC]
    cin = sum(segC)
    w = torch.randn(Cout, cin, k, k, k, generator=g, dtype=torch.float64) / (cin * k ** 3) ** 0.5
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    gy = torch.randn(B, Cout, D, H, W, generator=g, dtype=torch.float64).float().double()
    xr = [x.clone().requires_grad_(True) for x in xs]
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = ref_conv_block(xr, [norm] * len(segC), [None] * len(segC), wr, br, k, 1, 1, None)
    yr.backward(gy)
    xd = [cl(x).to(dev, torch.bfloat16).requires_grad_(True) for x in xs]
    wd, bd = w.float().to(dev).requires_grad_(True), b.float().to(dev).requires_grad_(True)
    segs = [ops.SegIn(x, stats=ops.instnorm_stats(x.detach()) if norm else None) for x in xd]
    y = ops.conv_block(segs, wd, bd, D, H, W, ksize=k, stride=1, pad_mode=1, out_dtype=torch.float32)
    assert y.dtype == torch.float32
    y.backward(cl(gy).to(dev, torch.float32))
    tol = TOL[torch.bfloat16]
    assert relerr(nc(y.detach()), yr) < tol
    assert relerr(wd.grad, wr.grad) < tol * 2 and relerr(bd.grad, br.grad) < tol * 2
    for i in range(len(segC)):
        assert relerr(nc(xd[i].grad.float()), xr[i].grad) < tol * 3


# 3x3x3 over two segments, and the shape of the fusion blocks' first conv (blocks.py:536): 1x1x1 over four modality segments
@pytest.mark.parametrize("ks,nseg,C,dims", [(3, 2, 8, (8, 8, 16)), (1, 4, 8, (8, 8, 16)), (1, 4, 16, (8, 8, 16)),
                                            # >= 32768 voxels: block-diagonal dgrad on the streaming pointwise kernel, finish-free
                                            (1, 4, 8, (32, 32, 48)), (1, 4, 16, (32, 32, 32)), (1, 2, 8, (32, 64, 32)),
                                            # >= 24 M input elements: the means from the weight gradient, summed over the passes
                                            (1, 4, 16, (64, 64, 64)), (1, 4, 8, (64, 96, 64))])
def test_conv_batched_passes(ks, nseg, C, dims):
    """SegIn.bmod: P masked passes over the same B samples in one launch == P separate launches, gradients summed"""
    from passion_amd import ops
    dev = _dev()
    B, P, Cout, (D, H, W) = 2, 3, C, dims
    g = torch.Generator().manual_seed(5)
    xs = [(torch.randn(B, D, H, W, C, generator=g) * 1.3).to(dev, torch.bfloat16) for _ in range(nseg)]
    w = (torch.randn(Cout, nseg * C, ks, ks, ks, generator=g) / (nseg * C * ks ** 3) ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    masks = (torch.rand(P * B, nseg, generator=g) > 0.35).to(dev)   # (P*B, nseg)
    masks[0] = True
    masks[4] = False
    gy = torch.randn(P * B, D, H, W, Cout, generator=g).to(dev, torch.bfloat16)
    stats = [ops.instnorm_stats(x) for x in xs]

    def run(batched):
        xl = [x.clone().requires_grad_(True) for x in xs]
        wl, bl = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if batched:
            segs = [ops.SegIn(x, stats=st, mask=masks, mask_col=i, bmod=B) for i, (x, st) in enumerate(zip(xl, stats))]
            y = ops.conv_block(segs, wl, bl, D, H, W, ksize=ks)
            y.backward(gy)
        else:
            ys = []
            for p_ in range(P):
                mk = masks[p_ * B:(p_ + 1) * B].contiguous()
                segs = [ops.SegIn(x, stats=st, mask=mk, mask_col=i) for i, (x, st) in enumerate(zip(xl, stats))]
                ys.append(ops.conv_block(segs, wl, bl, D, H, W, ksize=ks))
            y = torch.cat(ys, 0)
            y.backward(gy)
        return y.detach().float(), [x.grad.float() for x in xl], wl.grad, bl.grad

    ya, dxa, dwa, dba = run(True)
    yb, dxb, dwb, dbb = run(False)
    assert torch.equal(ya, yb)
    assert relerr(dwa, dwb) < 2e-3 and relerr(dba, dbb) < 2e-3
    for a, c in zip(dxa, dxb):
        assert relerr(a, c) < 2e-2     # bf16 per-pass gradients are summed in a different order


def test_conv_raw_input_segments():
    """first encoder conv: one modality of the NCDHW input = one C=1 segment, masked per sample"""
    from passion_amd import ops
    dev = _dev()
    B, S = 2, 8
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 4, S, S, S, generator=g)
    w = torch.randn(8, 1, 3, 3, 3, generator=g) / 5
    b = torch.randn(8, generator=g)
    mask = torch.tensor([[True, False, True, True], [False, True, True, False]])
    xd, md = x.to(dev), mask.to(dev)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    for m in range(4):
        seg = ops.SegIn(xd, mask=md, mask_col=m, C=1, offset=m * S ** 3, bstride=4 * S ** 3)
        y = ops.conv_block([seg], wd, bd, S, S, S, ksize=3, stride=1, pad_mode=1)
        xm = (x[:, m:m + 1] * mask[:, m].view(B, 1, 1, 1, 1)).double()
        yr = F.conv3d(F.pad(xm, (1,) * 6, mode="reflect"), w.double(), b.double())
        assert relerr(nc(y.detach()), yr) < 2e-4
    y.sum().backward()
    assert wd.grad is not None and torch.isfinite(wd.grad).all()


# one-launch path (small tensors, 1..4 blocks per sample) and the partial + finalize path (large ones)
@pytest.mark.parametrize("shape", [(2, 6, 10, 12, 16), (1, 8, 8, 8, 128), (2, 16, 16, 16, 64), (1, 32, 32, 32, 32), (1, 3, 5, 7, 4),
                                   (1, 64, 64, 48, 16), (2, 40, 40, 40, 8), (1, 4, 4, 4, 512)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_instnorm_stats(dtype, shape):
    from passion_amd import ops
    dev = _dev()
    C = shape[-1]
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(0)) * 3 + 5
    x = x + 0.05 * torch.arange(C).float()   # channels differ: a channel mix-up cannot hide
    x = x.to(dtype)
    st = ops.instnorm_stats(x.to(dev)).cpu().double()
    xf = x.double().reshape(shape[0], -1, C)
    mean, var = xf.mean(1), xf.var(1, unbiased=False)
    assert relerr(st[..., 0], mean) < 1e-5
    assert relerr(st[..., 1], 1 / torch.sqrt(var + 1e-5)) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,C,shape", [(2, 8, (4, 6, 8)), (2, 128, (2, 2, 2)), (4, 4, (4, 4, 4)), (8, 4, (2, 4, 2)),
                                       (16, 4, (2, 2, 2)), (2, 32, (3, 4, 40)), (2, 16, (5, 3, 24)),    # two / one vector per thread and row
                                       (2, 16, (45, 91, 6)), (2, 8, (64, 64, 5))])   # bf16: the 2 x 2-row-block adjoint (>= 2048 blocks, odd D and H)
def test_upsample(k, C, shape, dtype):
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, C, *shape, generator=g).to(dtype).double()
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=k, mode="trilinear", align_corners=True)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64).to(dtype).double()
    yr.backward(gy)
    xd = cl(x).to(dev, dtype).requires_grad_(True)
    y = ops.upsample(xd, k)
    y.backward(cl(gy).to(dev, dtype))
    tol = TOL[dtype]
    assert relerr(nc(y.detach().float()), yr) < tol
    assert relerr(nc(xd.grad.float()), xr.grad) < tol * 2


@pytest.mark.parametrize("B,Bt,dims,inv_temp", [(4, 1, (16, 16, 16), 1.0), (2, 2, (8, 12, 10), 1.0), (8, 2, (6, 6, 6), 0.25), (4, 1, (64, 64, 32), 1.0)])
def test_losses_straight_from_the_logits_equal_the_two_op_form(B, Bt, dims, inv_temp):
    """ops.softmax_wce_dice (decoder_sep's loss: the probabilities are never stored) against ops.softmax_classes -> ops.wce_dice:
    losses and the gradient wrt the logits bit for bit, with several predictions per label volume (Bt < B) and a temperature"""
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 31 + Bt)
    logits = (torch.randn(B, *dims, 4, generator=g) * 3.0).to(dev)
    lab = torch.randint(0, 4, (Bt, *dims), generator=g)
    tcl = F.one_hot(lab, 4).float().to(dev).contiguous()
    gout = torch.randn(B, 2, generator=g).to(dev)
    la = logits.clone().requires_grad_(True)
    oa = ops.wce_dice(ops.softmax_classes(la, inv_temp, "cl"), tcl, "cl")
    oa.backward(gout)
    lb = logits.clone().requires_grad_(True)
    ob = ops.softmax_wce_dice(lb, tcl, inv_temp)
    ob.backward(gout)
    assert torch.equal(oa, ob), (oa, ob)
    assert torch.equal(la.grad, lb.grad), float((la.grad - lb.grad).abs().max())
    assert float(lb.grad.abs().max()) > 0


def _ref_wce(p, t):
    B = p.shape[0]
    out = 0
    for i in range(4):
        w = 1.0 - t[:, i].sum((1, 2, 3)) / t.sum((1, 2, 3, 4))
        out = out - w.view(B, 1, 1, 1) * t[:, i] * torch.log(torch.clamp(p[:, i], 0.005, 1))
    return out.mean((1, 2, 3))


def _ref_dice(p, t):
    d = 0
    for i in range(4):
        num = (p[:, i] * t[:, i]).sum((1, 2, 3))
        d = d + 2.0 * num / (p[:, i].sum((1, 2, 3)) + t[:, i].sum((1, 2, 3)) + 1e-7)
    return 1.0 - d / 4


def _onehot(B, S, g):
    lab = torch.randint(0, 4, (B, S, S, S), generator=g)
    return F.one_hot(lab, 4).permute(0, 4, 1, 2, 3).double()


@pytest.mark.parametrize("layout", ["cl", "ncdhw"])
def test_softmax_wce_dice(layout):
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    B, S = 2, 12
    logits = (torch.randn(B, 4, S, S, S, generator=g) * 4).double()
    target = _onehot(B, S, g)
    lr = logits.clone().requires_grad_(True)
    pr = torch.softmax(lr / 2.0, 1)
    ce, dice = _ref_wce(pr, target), _ref_dice(pr, target)
    (ce * torch.tensor([1.0, 2.0]) + dice * torch.tensor([0.5, 3.0])).sum().backward()

    ld = cl(logits).float().to(dev).requires_grad_(True)
    tcl = ops.target_to_cl(target.to(dev))
    p = ops.softmax_classes(ld, 0.5, layout)
    out = ops.wce_dice(p, tcl, layout)
    (out[:, 0] * torch.tensor([1.0, 2.0], device=dev) + out[:, 1] * torch.tensor([0.5, 3.0], device=dev)).sum().backward()
    pn = p.detach() if layout == "ncdhw" else nc(p.detach())
    assert relerr(pn, pr.detach()) < 1e-5
    assert relerr(out[:, 0].detach(), ce.detach()) < 1e-5
    assert relerr(out[:, 1].detach(), dice.detach()) < 1e-5
    assert relerr(nc(ld.grad), lr.grad) < 1e-4


def test_kl():
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(10)
    B, S, T = 2, 10, 4.0
    ls = (torch.randn(B, 4, S, S, S, generator=g) * 6).double()
    lt = (torch.randn(B, 4, S, S, S, generator=g) * 6).double()
    lsr = ls.clone().requires_grad_(True)
    ps, pt = torch.softmax(lsr / T, 1), torch.softmax(lt / T, 1)
    psc, ptc = torch.clamp(ps, 0.005, 1), torch.clamp(pt, 0.005, 1)
    kl = (T * T * ptc * (torch.log(ptc) - torch.log(psc))).mean((1, 2, 3, 4))
    (kl * torch.tensor([1.0, 2.0])).sum().backward()
    lsd = cl(ls).float().to(dev).requires_grad_(True)
    ltd = cl(lt).float().to(dev)
    out = ops.kl_probs(ops.softmax_classes(lsd, 1 / T), ops.softmax_classes(ltd, 1 / T), T)
    (out[:, 0] * torch.tensor([1.0, 2.0], device=dev)).sum().backward()
    assert relerr(out[:, 0].detach(), kl.detach()) < 1e-5
    assert relerr(nc(lsd.grad), lsr.grad) < 1e-4


def _ref_proto(fs, ft, target):
    eps = 1e-5
    s, t = [], []
    for i in range(4):
        ti = target[:, i]
        if (ti.sum((-3, -2, -1)) > 0).all():
            den = ti[:, None].sum((-3, -2, -1)) + eps
            ps = (fs * ti[:, None]).sum((-3, -2, -1)) / den
            pt = (ft * ti[:, None]).sum((-3, -2, -1)) / den
            s.append(F.cosine_similarity(fs, ps[:, :, None, None, None], dim=1, eps=eps).unsqueeze(1))
            t.append(F.cosine_similarity(ft, pt[:, :, None, None, None], dim=1, eps=eps).unsqueeze(1))
    s, t = torch.cat(s, 1), torch.cat(t, 1)
    return ((s - t) ** 2).mean((1, 2, 3, 4)), torch.sqrt((s - t) ** 2).mean((1, 2, 3, 4))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop_class", [False, True])
@pytest.mark.parametrize("Cf", [8, 4])
def test_proto(dtype, drop_class, Cf):
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    B, S = 2, 8
    fs = torch.randn(B, Cf, S, S, S, generator=g).to(dtype).double()
    ft = torch.randn(B, Cf, S, S, S, generator=g).to(dtype).double()
    target = _onehot(B, S, g)
    if drop_class:  # class 3 absent from sample 1 -> skipped for the whole batch (criterions.py:157)
        t1 = target[1]
        t1[0] += t1[3]
        t1[3] = 0
    fr = fs.clone().requires_grad_(True)
    loss, dist = _ref_proto(fr, ft, target)
    (loss * torch.tensor([1.0, 2.0])).sum().backward()
    fd = cl(fs).to(dev, dtype).requires_grad_(True)
    lo, di = ops.proto_loss(fd, cl(ft).to(dev, dtype), ops.target_to_cl(target.to(dev)))
    (lo[:, 0] * torch.tensor([1.0, 2.0], device=dev)).sum().backward()
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert relerr(lo[:, 0].detach(), loss.detach()) < tol
    assert relerr(di[:, 0].detach(), dist.detach()) < tol
    assert relerr(nc(fd.grad.float()), fr.grad) < tol * 3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [8, 125, 256, 512, 2048])      # 256 / 512 / 1024 / 2048: register-resident softmax rows
def test_attention(N, dtype):
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(13)
    B, Hh, Dh = 2, 8, 64
    qkv = torch.randn(B, N, 3 * Hh * Dh, generator=g).to(dtype).double()
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.reshape(B, N, 3, Hh, Dh).permute(2, 0, 3, 1, 4)
    att = ((q @ k.transpose(-2, -1)) * Dh ** -0.5).softmax(-1)
    outr = (att @ v).transpose(1, 2).reshape(B, N, Hh * Dh)
    go = torch.randn(outr.shape, generator=g, dtype=torch.float64).to(dtype).double()
    outr.backward(go)
    qd = qkv.to(dev, dtype).requires_grad_(True)
    out = ops.attention(qd, Hh, Dh ** -0.5)
    out.backward(go.to(dev, dtype))
    tol = TOL[dtype]
    assert relerr(out.detach().float(), outr) < tol
    assert relerr(qd.grad.float(), qr.grad) < tol * 3


def test_layout_roundtrip():
    from passion_amd import ops
    dev = _dev()
    x = torch.randn(2, 4, 3, 5, 7).to(dev)
    assert torch.equal(ops.to_ncdhw(ops.to_channels_last(x)), x)
    y = ops.to_channels_last(x, torch.bfloat16)
    assert torch.equal(y, x.permute(0, 2, 3, 4, 1).contiguous().to(torch.bfloat16))


@pytest.mark.parametrize("amsgrad", [True, False])
def test_fused_adamw_matches_torch(amsgrad):
    """passion_amd.optim.FusedAdamW == torch.optim.AdamW (same state layout, interchangeable state_dict)"""
    from passion_amd.optim import FusedAdamW
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    shapes = [(8, 1, 3, 3, 3), (8,), (128, 128, 3, 3, 3), (1, 125, 512), (4097,), (3,), (512, 2048, 1, 1, 1)]
    pa = [torch.randn(s_, generator=g).to(dev).requires_grad_(True) for s_ in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = FusedAdamW(pa, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, amsgrad=amsgrad)
    v0 = [p._version for p in pa]
    ob = torch.optim.AdamW(pb, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, amsgrad=amsgrad)
    for it in range(4):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(dev) * (10.0 ** (it - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert relerr(a.detach(), b.detach()) < 2e-6
    assert all(p._version > v for p, v in zip(pa, v0)), "in-place update must bump the version (packed-weight caches key on it)"
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert set(sa["state"][k]) == set(sb["state"][k])
        assert relerr(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"]) < 2e-6
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
    ob.load_state_dict(sa)      # interchangeable


def test_dropout_mask_is_bernoulli_and_reproducible():
    """passion_dropout: keep rate 1 - p, survivors scaled by 1 / (1 - p), the mask is a pure function of (seed, index)
    -- the backward of the attention op applies it again to dprobs -- and different seeds give independent masks"""
    from ctypes import c_void_p
    from passion_amd import _lib as L
    dev = _dev()
    lib = L.load()
    n, p = 1_000_003, 0.1
    x = torch.rand(n, device=dev) + 0.5

    def run(src, seed):
        y = torch.empty_like(src)
        L.check(lib.passion_dropout(c_void_p(src.data_ptr()), c_void_p(y.data_ptr()), n, p, seed, L.stream_ptr()), "passion_dropout")
        return y

    y1, y2, y3 = run(x, 12345), run(x, 12345), run(x, 999)
    assert torch.equal(y1, y2)
    keep = y1 != 0
    assert abs(float(keep.float().mean()) - (1 - p)) < 3e-3
    assert torch.allclose(y1[keep], x[keep] / (1 - p), rtol=1e-6)
    both = float((keep & (y3 != 0)).float().mean())
    assert abs(both - (1 - p) ** 2) < 3e-3                      # independent masks
    g = run(torch.ones_like(x), 12345)                          # what backward multiplies dprobs with
    assert torch.equal(g != 0, keep)
    # no structure along the index: neighbouring elements (which share one hash value) are uncorrelated
    k = keep.float()
    c = float(((k[:-1] - (1 - p)) * (k[1:] - (1 - p))).mean()) / (p * (1 - p))
    assert abs(c) < 5e-3


def test_attention_dropout_backward_uses_the_forward_mask():
    from passion_amd import ops
    dev = _dev()
    torch.manual_seed(3)
    B, N, H, Dh = 1, 64, 2, 16
    qkv = torch.randn(B, N, 3 * H * Dh, device=dev, requires_grad=True)
    torch.manual_seed(11)
    out = ops.attention(qkv, H, Dh ** -0.5, 0.5)
    (g,) = torch.autograd.grad(out.sum(), qkv)
    # finite differences along a random direction see the same (fixed-seed) mask
    d = torch.randn_like(qkv) * 1e-3

    def f(t):
        torch.manual_seed(11)
        return float(ops.attention(t, H, Dh ** -0.5, 0.5).double().sum())

    num = (f(qkv.detach() + d) - f(qkv.detach() - d)) / 2
    ana = float((g * d).double().sum())
    assert abs(num - ana) < 2e-2 * max(abs(num), abs(ana), 1e-3)


MARCH_CASES = [
    # shapes the marching kernel (csrc/conv_march.hip) covers: 3x3x3 stride 1, <= 16 in / out channels, H % 8 == 0, W % 16 == 0
    ([8], 8, 3, 1, 1, (12, 24, 32), True, False, True),        # one block, 32-wide tiles, residual
    ([8, 8], 8, 3, 1, 1, (9, 16, 48), True, True, False),      # two segments, one absent in sample 0/1, 16-wide tiles
    ([16], 16, 3, 1, 0, (8, 24, 32), True, False, False),      # zero padding (forward only: its dgrad stays on the lean kernel)
    ([8], 8, 3, 1, 1, (6, 8, 16), False, False, False),        # not normalised: fold-only finish
    ([8, 8], 16, 3, 1, 1, (16, 16, 64), True, True, False),
    ([16], 8, 3, 1, 1, (40, 32, 32), True, False, False),      # several depth segments per tile
    ([8], 4, 3, 1, 1, (5, 8, 32), True, False, False),         # 4 output channels (RFNet-style narrow head input)
    ([8, 8, 8], 8, 3, 1, 1, (10, 16, 32), True, True, False),  # three segments: forward as NB = 4 with an empty block (RFNet's level-1 decoder concat)
    ([8, 8, 8], 16, 3, 1, 1, (8, 8, 16), True, False, False),
    ([16, 16], 8, 3, 1, 1, (9, 16, 32), True, True, False),    # input gradient with 32 rows: two row tiles in the marching dgrad
    ([32], 16, 3, 1, 1, (12, 8, 48), True, False, False),      # 32 rows over two K blocks (forward: lean kernel)
    ([8, 16], 8, 3, 1, 1, (6, 24, 16), True, True, False),     # 24 rows, segments of different width
]
MARCH_PATHS = {"march": {}, "march_one_segment": {"conv_target": 1}, "march_many_segments": {"conv_target": 100000},
               "march_off": {"conv_march": 0}}


@pytest.mark.parametrize("path", list(MARCH_PATHS))
@pytest.mark.parametrize("case", MARCH_CASES, ids=[str(i) for i in range(len(MARCH_CASES))])
def test_conv_march_kernel(case, path):
    from passion_amd import ops
    _dev()
    old = {k: ops.set_option(k, v) for k, v in MARCH_PATHS[path].items()}
    try:
        test_conv_block_fwd_bwd(case, torch.bfloat16)
    finally:
        for k, v in old.items():
            ops.set_option(k, v)


def test_conv_march_backward_is_bitwise_reproducible():
    """the marching dgrad leaves its InstanceNorm-backward sums in per-workgroup slots (fixed-order fold): no f32 atomics"""
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 24, 32, 64, 8, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(8, 8, 3, 3, 3, generator=g) * 0.1).to(dev).requires_grad_(True)
    gy = torch.randn(2, 24, 32, 64, 8, generator=g).to(dev, torch.bfloat16)
    outs = []
    for _ in range(3):
        xd = x.clone().requires_grad_(True)
        y = ops.conv_block([ops.SegIn(xd, stats=ops.instnorm_stats(xd.detach()))], w, None, 24, 32, 64, ksize=3, stride=1, pad_mode=1)
        w.grad = None
        y.backward(gy)
        outs.append((xd.grad.clone(), w.grad.clone()))
    assert all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
    assert all(torch.equal(outs[0][1], o[1]) for o in outs[1:])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("R,C", [(1280, 4096), (256, 512), (67, 130), (5, 2), (2560, 1536)])
def test_colsum_bias_gradient(R, C, dtype):
    from passion_amd import ops
    """passion_colsum (the bias gradient of the token GEMMs) against a float64 column sum, and bit-identical over repeats
    while a second stream keeps the GPU busy (torch's own sum is not, on this stack: tools/linear_bias_determinism.py)"""
    dev = torch.device("cuda:0")
    g = (torch.randn(R, C, device=dev, generator=torch.Generator(dev).manual_seed(R + C)) * 0.1).to(dtype)
    want = g.double().sum(0)
    got = ops.colsum(g)
    assert got.dtype == torch.float32 and got.shape == (C,)
    assert float((got.double() - want).abs().max()) <= 2e-6 * float(g.double().abs().sum(0).max()) + 1e-9
    side = torch.cuda.Stream()
    a = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16)
    for it in range(50):
        if it % 4 == 0:
            with torch.cuda.stream(side):
                a @ a
        assert torch.equal(ops.colsum(g), got)
    torch.cuda.synchronize()


def test_linear_fn_matches_torch_linear():
    from passion_amd import ops
    """ops.linear (hipBLASLt GEMMs + passion_colsum bias gradient) against torch's F.linear autograd, f32"""
    dev = torch.device("cuda:0")
    gen = torch.Generator(dev).manual_seed(3)
    x = torch.randn(3, 40, 64, device=dev, generator=gen, requires_grad=True)
    w = (torch.randn(96, 64, device=dev, generator=gen) * 0.1).requires_grad_(True)
    b = torch.randn(96, device=dev, generator=gen).requires_grad_(True)
    gy = torch.randn(3, 40, 96, device=dev, generator=gen)
    y = ops.linear(x, w, b, False)
    y.backward(gy)
    got = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    y2 = torch.nn.functional.linear(x, w, b)
    y2.backward(gy)
    for a_, b_ in zip(got, (y2.detach(), x.grad, w.grad, b.grad)):
        assert float((a_ - b_).abs().max()) <= 1e-4 * float(b_.abs().max())


FUSED_STATS_CASES = [
    # (segment channels, Cout, ksize, stride, pad, (D, H, W), dtype, residual, path the forward takes)
    ([8], 8, 3, 1, 1, (24, 32, 64), torch.bfloat16, True, "marching kernel epilogue"),
    ([8, 8], 16, 3, 1, 1, (16, 16, 32), torch.bfloat16, False, "marching kernel epilogue, two segments"),
    ([16], 16, 3, 1, 0, (40, 32, 32), torch.bfloat16, False, "marching kernel, several depth segments"),
    ([32], 32, 3, 1, 1, (16, 16, 16), torch.bfloat16, True, "lean kernel epilogue"),
    ([64], 64, 3, 1, 1, (8, 8, 8), torch.bfloat16, False, "lean kernel epilogue, 8^3"),
    ([32, 32], 32, 1, 1, 0, (16, 16, 16), torch.bfloat16, False, "lean 1x1x1"),
    ([128], 128, 3, 1, 1, (8, 8, 8), torch.bfloat16, False, "lean, several row-tile groups"),
    ([8], 16, 3, 2, 1, (16, 16, 16), torch.bfloat16, False, "stride 2: general bf16 kernel epilogue"),
    ([16], 32, 3, 2, 1, (64, 64, 64), torch.bfloat16, False, "stride 2, 1024 bricks per sample: two-level fold"),
    ([8], 8, 3, 2, 1, (10, 12, 20), torch.bfloat16, False, "stride 2, ragged bricks"),
    ([8], 8, 3, 1, 1, (10, 12, 20), torch.bfloat16, False, "ragged bricks: statistics kernels after the conv"),
    ([8], 8, 3, 1, 1, (8, 8, 16), torch.float32, False, "f32 storage: statistics kernels after the conv"),
    ([16], 16, 1, 1, 0, (32, 32, 32), torch.bfloat16, False, "pointwise streaming kernel (conv_pw.hip), one K step"),
    ([16, 16, 16, 16], 32, 1, 1, 0, (33, 32, 40), torch.bfloat16, False, "pointwise streaming kernel, two K steps x two row tiles, ragged"),
]


@pytest.mark.parametrize("case", FUSED_STATS_CASES, ids=[c[-1] for c in FUSED_STATS_CASES])
def test_conv_forward_leaves_the_statistics_of_its_output(case):
    """passion_conv3d_fwd_stats: y bit-identical to passion_conv3d_fwd, statistics = passion_instnorm_stats(y) (the fused
    epilogue sums the stored bf16 values in f32 per workgroup, the separate kernels per thread: agreement to f32 rounding),
    bit-identical over repeats, also into a preallocated slot and for a batch of 3"""
    from passion_amd import ops
    dev = _dev()
    segc, Cout, ks, stride, pad, (D, H, W), dtype, with_res, _ = case
    B = 3
    g = torch.Generator().manual_seed(sum(segc) + Cout + D)
    xs = [(torch.randn(B, D, H, W, c, generator=g) * 2 + 0.7).to(dev, dtype) for c in segc]
    w = (torch.randn(Cout, sum(segc), ks, ks, ks, generator=g) * 0.2).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    Do, Ho, Wo = [(s + 2 * (ks // 2) - ks) // stride + 1 for s in (D, H, W)]
    res = (torch.randn(B, Do, Ho, Wo, Cout, generator=g).to(dev, dtype)) if with_res else None
    segs = lambda: [ops.SegIn(x, stats=ops.instnorm_stats(x)) for x in xs]
    kw = dict(ksize=ks, stride=stride, pad_mode=pad, residual=res)
    y0 = ops.conv_block(segs(), w, bias, D, H, W, **kw)
    y1, st1 = ops.conv_block(segs(), w, bias, D, H, W, want_stats=True, **kw)
    assert torch.equal(y0, y1)
    want = ops.instnorm_stats(y0)
    assert st1.shape == (B, Cout, 2) and st1.dtype == torch.float32
    std = 1.0 / want[..., 1]
    assert float(((st1[..., 0] - want[..., 0]).abs() / (std + want[..., 0].abs())).max()) < 2e-5
    assert float(((st1[..., 1] - want[..., 1]).abs() / want[..., 1]).max()) < 1e-4
    slot = torch.full((2, B, Cout, 2), -7.0, device=dev)
    for _ in range(3):
        y2, st2 = ops.conv_block(segs(), w, bias, D, H, W, want_stats=True, stats_out=slot[1], **kw)
        assert st2.data_ptr() == slot[1].data_ptr() and torch.equal(st2, st1) and torch.equal(y2, y0)
    assert float(slot[0].min()) == -7.0 and float(slot[0].max()) == -7.0
    # the statistics are plain data for autograd: the backward of y is unchanged
    xg = [x.clone().requires_grad_(True) for x in xs]
    wg = w.clone().requires_grad_(True)
    y3, _ = ops.conv_block([ops.SegIn(x, stats=ops.instnorm_stats(x.detach())) for x in xg], wg, bias, D, H, W, want_stats=True, **kw)
    y3.float().square().sum().backward()
    assert wg.grad is not None and all(x.grad is not None for x in xg)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("shape", [(2, 8, 8, 8, 16), (1, 16, 16, 16, 8), (3, 4, 6, 5, 64), (1, 32, 32, 32, 16), (2, 3, 3, 3, 4)])
def test_upsample_forward_leaves_the_statistics_of_its_output(shape, dtype):
    """passion_upsample_fwd_stats: y bit-identical to passion_upsample_fwd, statistics = passion_instnorm_stats(y), repeatable"""
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 1.5 - 0.4).to(dev, dtype)
    y0 = ops.upsample(x, 2)
    y1, st = ops.upsample_with_stats(x, 2)
    assert torch.equal(y0, y1)
    want = ops.instnorm_stats(y0)
    std = 1.0 / want[..., 1]
    assert float(((st[..., 0] - want[..., 0]).abs() / (std + want[..., 0].abs())).max()) < 2e-5
    assert float(((st[..., 1] - want[..., 1]).abs() / want[..., 1]).max()) < 1e-4
    for _ in range(3):
        y2, st2 = ops.upsample_with_stats(x, 2)
        assert torch.equal(st2, st) and torch.equal(y2, y0)
    xg = x.clone().requires_grad_(True)
    y3, _ = ops.upsample_with_stats(xg, 2)
    y3.float().sum().backward()
    xr = x.clone().requires_grad_(True)
    ops.upsample(xr, 2).float().sum().backward()
    assert torch.equal(xg.grad, xr.grad)


GROUPED_CASES = [
    # (segment channels, Cout, ksize, stride, pad, (D, H, W), dtype, residual, normalised input, per-sample modality mask)
    ([8], 8, 3, 1, 1, (16, 16, 32), torch.bfloat16, True, True, True),        # marching kernels
    ([16], 16, 3, 1, 1, (12, 24, 16), torch.bfloat16, False, True, False),
    ([32], 32, 3, 1, 1, (8, 16, 16), torch.bfloat16, True, True, True),       # lean kernels
    ([64], 64, 3, 1, 1, (8, 8, 8), torch.bfloat16, False, True, False),
    ([8], 16, 3, 2, 1, (16, 16, 16), torch.bfloat16, False, True, True),      # stride 2: general forward / wgrad, lean dgrad
    ([1], 8, 3, 1, 1, (8, 16, 16), torch.bfloat16, False, False, True),       # the first conv of an encoder (Cin = 1)
    ([16, 16], 16, 3, 1, 1, (8, 8, 16), torch.bfloat16, False, True, False),  # two segments
    ([8], 8, 3, 1, 1, (8, 8, 16), torch.float32, True, True, True),           # f32 storage: set by set
    ([32], 64, 3, 2, 1, (16, 32, 32), torch.bfloat16, False, True, True),     # stride 2, several bricks per slab slot: the grouped general wgrad
    ([16], 32, 3, 2, 1, (24, 16, 16), torch.bfloat16, False, True, False),    # stride 2, ragged brick count per sample
    ([1], 8, 3, 1, 1, (16, 16, 32), torch.bfloat16, False, False, True),      # the first conv on the matrix-core weight gradient (conv_c1_mfma.hip)
]


@pytest.mark.parametrize("bg", [1, 2], ids=["one_sample_per_set", "two_samples_per_set"])
@pytest.mark.parametrize("case", GROUPED_CASES, ids=[str(i) for i in range(len(GROUPED_CASES))])
def test_grouped_conv_equals_the_four_convs(case, bg):
    """ops.conv_block_grouped (four weight sets over one stacked batch, one launch per kernel) against four ops.conv_block
    calls on the four slices: outputs, statistics and input gradients bit for bit (the same kernels, the same order of sums);
    weight gradients to f32 rounding (the grouped fold adds the same slots but in sample order)"""
    from passion_amd import ops
    dev = _dev()
    segc, Cout, ks, stride, pad, (D, H, W), dtype, with_res, norm, with_mask = case
    G, B = 4, 4 * bg
    g = torch.Generator().manual_seed(11 + sum(segc) + Cout + D + bg)
    xs = [(torch.randn(B, D, H, W, c, generator=g) * 1.5 + 0.3).to(dev, dtype) for c in segc]
    ws = [(torch.randn(Cout, sum(segc), ks, ks, ks, generator=g) * 0.2).to(dev).requires_grad_(True) for _ in range(G)]
    bs = [torch.randn(Cout, generator=g).to(dev).requires_grad_(True) for _ in range(G)]
    Do, Ho, Wo = [(s + 2 * (ks // 2) - ks) // stride + 1 for s in (D, H, W)]
    res = torch.randn(B, Do, Ho, Wo, Cout, generator=g).to(dev, dtype) if with_res else None
    mask = (torch.rand(B, 1, generator=g) > 0.3).to(dev) if with_mask else None
    gy = torch.randn(B, Do, Ho, Wo, Cout, generator=g).to(dev, dtype)
    kw = dict(ksize=ks, stride=stride, pad_mode=pad)

    def run(grouped):
        xg = [x.clone().requires_grad_(sum(segc) % 4 == 0) for x in xs]      # the raw input volume (Cin = 1) takes no gradient
        rg = res.clone().requires_grad_(True) if with_res else None
        for t in ws + bs:
            t.grad = None
        sts = [ops.instnorm_stats(x.detach()) if norm else None for x in xg]
        if grouped:
            segs = [ops.SegIn(x, stats=s, mask=mask, mask_col=0) for x, s in zip(xg, sts)]
            y, st = ops.conv_block_grouped(segs, ws, bs, D, H, W, residual=rg, want_stats=True, **kw)
        else:
            ys, stl = [], []
            for q in range(G):
                sl = slice(q * bg, (q + 1) * bg)
                segs = [ops.SegIn(x[sl], stats=None if s is None else s[sl], mask=None if mask is None else mask[sl], mask_col=0)
                        for x, s in zip(xg, sts)]
                yq, sq = ops.conv_block(segs, ws[q], bs[q], D, H, W, residual=None if rg is None else rg[sl], want_stats=True, **kw)
                ys.append(yq)
                stl.append(sq)
            y, st = torch.cat(ys, 0), torch.cat(stl, 0)
        y.backward(gy)
        return (y.detach(), st, [x.grad for x in xg], None if rg is None else rg.grad, [w.grad.clone() for w in ws], [b.grad.clone() for b in bs])

    yg, stg, dxg, drg, dwg, dbg = run(True)
    yr, str_, dxr, drr, dwr, dbr = run(False)
    assert torch.equal(yg, yr)
    assert torch.equal(stg, str_)
    for a, b in zip(dxg, dxr):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0
        elif dtype == torch.float32:     # the f32 kernels add the InstanceNorm-backward sums with f32 atomics: order-dependent
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
        else:
            assert torch.equal(a, b)
    if with_res:
        assert torch.equal(drg, drr)
    # (the grouped launch and a single set may take different weight-gradient kernels -- wgrad_march16_kernel wants >= 8 (sample, channel
    # block) pairs -- and the marching kernels normalise as fma(x, rstd, -mean rstd), the lean one as (x - mean) rstd: a last-bit
    # difference of x_hat before its bf16 rounding moves a few operands by one bf16 ulp, 2e-4 of the largest weight gradient measured)
    for a, b in zip(dwg + dbg, dwr + dbr):
        assert float((a - b).abs().max()) <= 5e-4 * float(b.abs().max()) + 1e-6


WGRAD_MARCH_CASES = [
    # 3x3x3 stride 1 with 8 output channels on (H % 8, W % 16) == 0 volumes: the marching weight-gradient kernel
    ([8], 8, 3, 1, 1, (12, 16, 32), True, False, False),
    ([8, 8], 8, 3, 1, 0, (9, 8, 16), True, True, False),        # zero padding, masked segments, ragged D split
    ([16], 8, 3, 1, 1, (20, 24, 48), False, False, False),      # raw input (no InstanceNorm in the loader)
    ([8], 8, 3, 1, 1, (40, 16, 16), True, False, True),         # several D segments per tile
    ([32], 8, 3, 1, 0, (8, 128, 256), True, False, False),      # fewer workgroups per sample than tiles: two units per workgroup
]


@pytest.mark.parametrize("case", WGRAD_MARCH_CASES, ids=[str(i) for i in range(len(WGRAD_MARCH_CASES))])
def test_wgrad_march(case):
    """conv_wgrad_march.hip against the f64 reference (forward / dgrad ride along), and that it is the kernel that ran"""
    from passion_amd import ops
    ops.prof_read()
    ops.prof_enable(1 << 3)
    try:
        test_conv_block_fwd_bwd(case, torch.bfloat16)
    finally:
        ops.prof_enable(0)
    names = {r[4] for r in ops.prof_read()}
    assert "wgrad_march_kernel<0>" in names, names


def test_wgrad_march_equals_lean_kernel_sums():
    """same launch through wgrad_lean_kernel and wgrad_march_kernel: both sum the same bf16 products in f32, only the order differs"""
    from passion_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, C, D, H, W = 2, 8, 16, 32, 32
    x = torch.randn(B, D, H, W, C, generator=g).to(dev, torch.bfloat16)
    gy = torch.randn(B, D, H, W, 8, generator=g).to(dev, torch.bfloat16)
    w0 = (torch.randn(8, C, 3, 3, 3, generator=g) / 15).to(dev)
    b0 = torch.randn(8, generator=g).to(dev)
    out = {}
    for march in (1, 0):
        old = ops.set_option("wgrad_march", march)
        try:
            xd = x.clone().requires_grad_(True)
            wd, bd = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            y = ops.conv_block([ops.SegIn(xd, stats=ops.instnorm_stats(xd.detach()))], wd, bd, D, H, W, ksize=3, stride=1, pad_mode=1)
            y.backward(gy)
            torch.cuda.synchronize()
            out[march] = (wd.grad.clone(), bd.grad.clone())
        finally:
            ops.set_option("wgrad_march", old)
    assert relerr(out[1][0], out[0][0].double()) < 1e-5
    assert relerr(out[1][1], out[0][1].double()) < 1e-5


@pytest.mark.parametrize("pad_mode", [1, 0], ids=["reflect", "zeros"])
@pytest.mark.parametrize("B,shape", [(2, (32, 32, 64)), (1, (5, 32, 16)), (3, (20, 16, 32)), (2, (9, 24, 128)), (1, (33, 8, 32))])
def test_first_layer_weight_gradient(B, shape, pad_mode):
    """wgrad_c1m_kernel (conv_c1_mfma.hip: one raw input channel, 8 outputs, bf16, on the matrix cores: reflect padding, W in {32, 64, 128},
    H % 8 == 0) and wgrad_c1_kernel (conv_wgrad_lean.hip: every other shape) against autograd in float64 on the same
    bf16-rounded operands; the first case has more slab slots (512 bricks) than summing workgroups: the spare slots must
    be flushed with zeros; a masked sample contributes nothing to dW but its dY still counts in the bias gradient"""
    from passion_amd import ops
    dev = _dev()
    D, H, W = shape
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(B, D, H, W, 1, generator=g) * 1.3 + 0.2).to(torch.bfloat16)
    w = torch.randn(8, 1, 3, 3, 3, generator=g) / 4
    b = torch.randn(8, generator=g)
    mask = torch.ones(B, 1, dtype=torch.bool)
    if B > 1:
        mask[1, 0] = False
    gy = torch.randn(B, D, H, W, 8, generator=g).to(torch.bfloat16)
    wr = w.to(torch.bfloat16).double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    xm = (x.double() * mask.view(-1, 1, 1, 1, 1)).permute(0, 4, 1, 2, 3)
    yr = F.conv3d(F.pad(xm, (1,) * 6, mode="reflect" if pad_mode == 1 else "constant"), wr, br)
    yr.backward(gy.double().permute(0, 4, 1, 2, 3))
    ops.prof_read()
    ops.prof_enable(15)
    try:
        wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        y = ops.conv_block([ops.SegIn(x.to(dev), mask=mask.to(dev), mask_col=0)], wd, bd, D, H, W, ksize=3, stride=1, pad_mode=pad_mode)
        y.backward(gy.to(dev))
        torch.cuda.synchronize()
    finally:
        ops.prof_enable(0)
    names = [r[4] for r in ops.prof_read()]
    mfma = pad_mode == 1 and W in (32, 64, 128) and H % 8 == 0
    assert (f"wgrad_c1m_kernel<{W}>" if mfma else "wgrad_c1_kernel") in names, names
    assert relerr(wd.grad, wr.grad) < 1e-5
    assert relerr(bd.grad, br.grad) < 1e-5
    first = wd.grad.clone()
    wd.grad = None
    bd.grad = None
    y = ops.conv_block([ops.SegIn(x.to(dev), mask=mask.to(dev), mask_col=0)], wd, bd, D, H, W, ksize=3, stride=1, pad_mode=pad_mode)
    y.backward(gy.to(dev))
    assert torch.equal(first, wd.grad)                     # fixed-order fold: bit-reproducible


def test_bf16_only_weight_pack():
    """ops.set_pack_f32(False) (what a bf16 model's forward says): a re-pack writes the two bf16 operand images only; a bf16 conv gives the
    same result bit for bit; an f32 conv on the same weight afterwards gets its f32 images back by itself (the image set is part of the
    pack's cache key); a bf16 conv that the library runs on its f32-image kernel (ragged concat) switches the process back for good;
    and the library refuses to run an f32-image kernel while it is told not to pack those images"""
    from passion_amd import ops, _lib as L
    dev = _dev()
    g = torch.Generator().manual_seed(41)
    D, H, W = 8, 16, 32
    x = torch.randn(2, D, H, W, 16, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(16, 16, 3, 3, 3, generator=g) / 20).to(dev)
    b = torch.randn(16, generator=g).to(dev)
    state = (ops._PACK_F32[0], ops._STICKY_F32[0])
    try:
        ops._STICKY_F32[0] = False
        ops.set_pack_f32(True)
        cache = {}
        y_all = ops.conv_block([ops.SegIn(x, stats=ops.instnorm_stats(x))], w, b, D, H, W, pack_cache=cache)
        ops.set_pack_f32(False)
        assert not ops._PACK_F32[0]
        y_bf = ops.conv_block([ops.SegIn(x, stats=ops.instnorm_stats(x))], w, b, D, H, W, pack_cache=cache)
        assert cache["key"][2] is False and torch.equal(y_all, y_bf)
        # the guard: an f32 conv straight through the C ABI while the option is off
        xf = x.float()
        with pytest.raises(RuntimeError, match="-1024"):
            d = ops._ConvCfg()
            d.segs, d.D, d.H, d.W, d.ksize, d.stride, d.pad_mode, d.slope, d.B, d.out, d.out_dtype = \
                [ops.SegIn(xf)], D, H, W, 3, 1, L.PAD_REFLECT, 0.2, 2, None, torch.float32
            ops.ConvBlockFn.apply(d, w, b, None, cache["packed"], xf)
        # ... and through ops.conv_block: the f32 images come back by themselves
        y32 = ops.conv_block([ops.SegIn(xf)], w, b, D, H, W, pack_cache=cache)
        assert ops._PACK_F32[0] and cache["key"][2] is True and not ops._STICKY_F32[0]
        ref = F.conv3d(F.pad(xf.permute(0, 4, 1, 2, 3).double(), (1,) * 6, mode="reflect"), w.double(), b.double())
        assert relerr(nc(y32), ref.cpu()) < TOL[torch.float32]
        # a ragged concat in bf16: the general kernel reads the f32 images -- all four from now on
        ops.set_pack_f32(False)
        x4 = torch.randn(2, D, H, W, 4, generator=g).to(torch.bfloat16).to(dev)
        w2 = (torch.randn(8, 20, 3, 3, 3, generator=g) / 20).to(dev)
        y_r = ops.conv_block([ops.SegIn(x), ops.SegIn(x4)], w2, None, D, H, W)
        assert ops._STICKY_F32[0] and ops._PACK_F32[0] and bool(torch.isfinite(y_r.float()).all())
        ops.set_pack_f32(False)
        assert ops._PACK_F32[0]                  # sticky
    finally:
        ops._STICKY_F32[0] = state[1]
        ops.set_pack_f32(state[0])


@pytest.mark.parametrize("C,shape", [(32, (8, 8, 16)), (16, (16, 32, 32)), (8, (24, 32, 64)), (8, (64, 128, 128)), (16, (5, 6, 7))])
def test_residual_gradient_joins_inside_the_conv_backward(C, shape):
    """x + c3(c2(x)) (the encoders' residual blocks, reference mmformer.py:34-38) with c2 handing x on to c3's residual input
    (conv_block_grouped(pass_x=True), passion_seg_t.dx_add): the gradient of x -- the identity branch's dy plus c2's input gradient --
    against the same block with autograd summing the two (bf16: the joined form rounds once instead of twice), and the weight
    gradients bit for bit; the shapes take the padded finish (levels 2..5), the final-form kernels + the library's own add (level 1)
    and the general kernels"""
    from passion_amd import ops
    dev = _dev()
    D, H, W = shape
    B, G = 4, 4
    g = torch.Generator().manual_seed(31)
    x0 = (torch.randn(B, D, H, W, C, generator=g) * 1.2 + 0.1).to(torch.bfloat16).to(dev)
    ws = [[(torch.randn(C, C, 3, 3, 3, generator=g) / (C * 27) ** 0.5).to(dev) for _ in range(G)] for _ in range(2)]
    bs = [[torch.randn(C, generator=g).to(dev) for _ in range(G)] for _ in range(2)]
    gy = torch.randn(B, D, H, W, C, generator=g).to(torch.bfloat16).to(dev)

    def run(joined):
        x = x0.clone().requires_grad_(True)
        xin = x * 1.0                                                # (a non-leaf input, as an encoder level's)
        w2, w3 = [w.clone().requires_grad_(True) for w in ws[0]], [w.clone().requires_grad_(True) for w in ws[1]]
        b2, b3 = [b.clone().requires_grad_(True) for b in bs[0]], [b.clone().requires_grad_(True) for b in bs[1]]
        seg = ops.SegIn(xin, stats=ops.instnorm_stats(xin.detach()))
        if joined:
            y2, st2, xres = ops.conv_block_grouped([seg], w2, b2, D, H, W, want_stats=True, pass_x=True)
        else:
            (y2, st2), xres = ops.conv_block_grouped([seg], w2, b2, D, H, W, want_stats=True), xin
        y = ops.conv_block_grouped([ops.SegIn(y2, stats=st2)], w3, b3, D, H, W, residual=xres)
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach(), x.grad, [w.grad for w in w2 + w3], [b.grad for b in b2 + b3]

    y1, dx1, dw1, db1 = run(True)
    y0, dx0, dw0, db0 = run(False)
    assert torch.equal(y1, y0)
    for a, b in zip(dw1 + db1, dw0 + db0):
        assert torch.equal(a, b)
    d = (dx1.float() - dx0.float()).abs()
    scale = dx0.float().abs().mean()
    assert float(d.max()) <= 2 ** -6 * float(dx0.float().abs().max()) and float(d.mean()) < 4e-3 * float(scale), (float(d.max()), float(d.mean()), float(scale))
    # ... and against f32 arithmetic on the unjoined pieces: the joined sum is the closer one
    assert relerr(dx1.float(), dx0.double()) < 1e-2

    # two further readers (a level's feature: decoder_sep and the fusion blocks): pass_x = 2, both gradients join in one backward call
    def run2(joined):
        x = x0.clone().requires_grad_(True)
        xin = x * 1.0
        w2, b2 = [w.clone().requires_grad_(True) for w in ws[0]], [b.clone().requires_grad_(True) for b in bs[0]]
        seg = ops.SegIn(xin, stats=ops.instnorm_stats(xin.detach()))
        if joined:
            y2, xa, xb = ops.conv_block_grouped([seg], w2, b2, D, H, W, pass_x=2)
        else:
            y2, xa, xb = ops.conv_block_grouped([seg], w2, b2, D, H, W), xin, xin
        return y2, xa, xb, x

    y2, xa, xb, x = run2(True)
    torch.autograd.backward([y2, xa, xb], [gy, gy * 0.5, gy * 0.25])
    torch.cuda.synchronize()
    dxj = x.grad.clone()
    y2, xa, xb, x = run2(False)
    torch.autograd.backward([y2, xa, xb], [gy, gy * 0.5, gy * 0.25])
    torch.cuda.synchronize()
    d = (dxj.float() - x.grad.float()).abs()
    assert float(d.max()) <= 2 ** -5 * float(x.grad.float().abs().max()) and float(d.mean()) < 6e-3 * float(x.grad.float().abs().mean())


@pytest.mark.parametrize("norm", [True, False], ids=["prenorm", "raw"])
@pytest.mark.parametrize("pad_mode", [1, 0], ids=["reflect", "zeros"])
@pytest.mark.parametrize("shape", [(32, 128, 128), (16, 128, 256), (26, 128, 160)])
def test_first_stride2_layer_weight_gradient(shape, pad_mode, norm):
    """wgrad_s2m_kernel (conv_s2_wgrad.hip: 3x3x3, stride 2, 8 -> 16, bf16, launches of >= 2^18 output voxels): weight and bias gradient
    against autograd in float64, against the brick kernel it replaces (the same bf16 products, another order of the f32 sums), a masked
    sample, bit-reproducible, and a grouped launch (four weight sets over the stacked batch) against the sets one by one"""
    from passion_amd import ops
    dev = _dev()
    D, H, W = shape
    B = 4
    g = torch.Generator().manual_seed(29)
    x = (torch.randn(B, 8, D, H, W, generator=g, dtype=torch.float64) * 1.4 + 0.25).to(torch.bfloat16).double()
    ws = [torch.randn(16, 8, 3, 3, 3, generator=g, dtype=torch.float64) / (8 * 27) ** 0.5 for _ in range(4)]
    bs = [torch.randn(16, generator=g, dtype=torch.float64) for _ in range(4)]
    mask = torch.tensor([[True], [False], [True], [True]])
    gy = torch.randn(B, 16, D // 2, H // 2, W // 2, generator=g, dtype=torch.float64).to(torch.bfloat16).double()
    wr, br = ws[0].to(torch.bfloat16).double().requires_grad_(True), bs[0].clone().requires_grad_(True)
    ref_conv_block([x], [norm], [mask[:, 0].double()], wr, br, 3, 2, pad_mode, None).backward(gy)
    xd, md, gyd = cl(x).to(dev, torch.bfloat16), mask.to(dev), cl(gy).to(dev, torch.bfloat16)

    def seg(sl=slice(None)):
        xs = xd[sl]
        return ops.SegIn(xs, stats=ops.instnorm_stats(xs) if norm else None, mask=md[sl], mask_col=0)

    def run(wi, bi, sl=slice(None)):
        wd, bd = wi.float().to(dev).requires_grad_(True), bi.float().to(dev).requires_grad_(True)
        y = ops.conv_block([seg(sl)], wd, bd, D, H, W, ksize=3, stride=2, pad_mode=pad_mode)
        y.backward(gyd[sl])
        torch.cuda.synchronize()
        return wd.grad.clone(), bd.grad.clone()

    ops.prof_read()
    ops.prof_enable(15)
    try:
        dW1, db1 = run(ws[0], bs[0])
    finally:
        ops.prof_enable(0)
    names = [r[4] for r in ops.prof_read()]
    assert "wgrad_s2m_kernel" in names, names
    tol = TOL[torch.bfloat16] if norm else 1e-5           # (raw input: the bf16 operands are the reference's own)
    assert relerr(dW1, wr.grad) < tol, relerr(dW1, wr.grad)
    assert relerr(db1, br.grad) < 1e-5
    dW2, db2 = run(ws[0], bs[0])
    assert torch.equal(dW1, dW2) and torch.equal(db1, db2)  # fixed-order fold: bit-reproducible
    old = ops.set_option("wgrad_s2m", 0)
    try:
        dW0, db0 = run(ws[0], bs[0])
    finally:
        ops.set_option("wgrad_s2m", old)
    assert relerr(dW1, dW0.double()) < 1e-4 and relerr(db1, db0.double()) < 1e-5
    # grouped: sample g under weight set g
    wg = [w.float().to(dev).requires_grad_(True) for w in ws]
    bg = [b.float().to(dev).requires_grad_(True) for b in bs]
    ops.prof_enable(15)
    try:
        yg = ops.conv_block_grouped([seg()], wg, bg, D, H, W, ksize=3, stride=2, pad_mode=pad_mode)
        yg.backward(gyd)
        torch.cuda.synchronize()
    finally:
        ops.prof_enable(0)
    assert "wgrad_s2m_kernel" in [r[4] for r in ops.prof_read()]
    for gi in range(4):
        dWs, dbs = run(ws[gi], bs[gi], slice(gi, gi + 1))     # (one sample: below the kernel's launch size, the brick kernel)
        assert relerr(wg[gi].grad, dWs.double()) < 1e-4, gi
        assert relerr(bg[gi].grad, dbs.double()) < 1e-5, gi
    assert float(wg[1].grad.abs().max()) == 0.0              # absent modality: no weight gradient, the bias gradient stays


@pytest.mark.parametrize("norm", [True, False], ids=["prenorm", "raw"])
@pytest.mark.parametrize("pad_mode", [1, 0], ids=["reflect", "zeros"])
@pytest.mark.parametrize("shape", [(64, 128, 128), (16, 128, 256), (22, 48, 32), (64, 64, 64), (6, 16, 32), (16, 12, 32)])
def test_first_stride2_layer_forward_kernel(shape, pad_mode, norm):
    """conv_s2_fwd.hip (3x3x3, stride 2, 8 -> 16, bf16: the marching forward of e2_c1, reference mmformer.py:40-45): against the f64
    reference, against the brick kernel it replaces, statistics of the output from the same launch, masked samples, plain and
    grouped (four weight sets over one stacked batch); shapes it does not take (launches of fewer than 256 workgroups, Ho % 8) stay on the brick kernel"""
    from passion_amd import ops
    dev = _dev()
    D, H, W = shape
    B = 4
    g = torch.Generator().manual_seed(23)
    x = (torch.randn(B, 8, D, H, W, generator=g, dtype=torch.float64) * 1.4 + 0.25).to(torch.bfloat16).double()
    ws = [torch.randn(16, 8, 3, 3, 3, generator=g, dtype=torch.float64) / (8 * 27) ** 0.5 for _ in range(4)]
    bs = [torch.randn(16, generator=g, dtype=torch.float64) for _ in range(4)]
    mask = torch.tensor([[True], [False], [True], [True]])
    xd, md = cl(x).to(dev, torch.bfloat16), mask.to(dev)

    def ref(wi, bi, samples):
        return ref_conv_block([x[samples]], [norm], [mask[samples, 0].double()], wi.to(torch.bfloat16).double(), bi, 3, 2, pad_mode, None)

    def seg():
        return ops.SegIn(xd, stats=ops.instnorm_stats(xd) if norm else None, mask=md, mask_col=0)

    ops.prof_read()
    ops.prof_enable(1)
    try:
        wd, bd = ws[0].float().to(dev), bs[0].float().to(dev)
        y, st = ops.conv_block([seg()], wd, bd, D, H, W, ksize=3, stride=2, pad_mode=pad_mode, want_stats=True)
        wg, bg = [w.float().to(dev) for w in ws], [b.float().to(dev) for b in bs]
        yg, stg = ops.conv_block_grouped([seg()], wg, bg, D, H, W, ksize=3, stride=2, pad_mode=pad_mode, want_stats=True)
        torch.cuda.synchronize()
    finally:
        ops.prof_enable(0)
    names = [r[4] for r in ops.prof_read()]
    Do, tiles = D // 2, (H // 16) * (W // 32)
    marching = (H // 2) % 8 == 0 and (W // 2) % 16 == 0 and Do >= 4 and B * tiles * max(1, min((1024 + B * tiles - 1) // (B * tiles), Do // 8)) >= 256
    assert names.count("conv_s2f_kernel") == (2 if marching else 0), names
    tol = TOL[torch.bfloat16]
    assert relerr(nc(y.float()), ref(ws[0], bs[0], slice(0, B))) < tol
    for gi in range(4):
        assert relerr(nc(yg[gi:gi + 1].float()), ref(ws[gi], bs[gi], slice(gi, gi + 1))) < tol
    assert float(y[1].abs().max()) == float(bs[0].to(torch.bfloat16).abs().max())      # absent modality: the bias alone
    for yy, ss in ((y, st), (yg, stg)):
        want = ops.instnorm_stats(yy)
        std = 1.0 / want[..., 1]
        assert float(((ss[..., 0] - want[..., 0]).abs() / (std + want[..., 0].abs())).max()) < 2e-5
        assert float(((ss[..., 1] - want[..., 1]).abs() / want[..., 1]).max()) < 1e-4
    old = ops.set_option("conv_s2f", 0)
    try:
        y0 = ops.conv_block([seg()], wd, bd, D, H, W, ksize=3, stride=2, pad_mode=pad_mode)
    finally:
        ops.set_option("conv_s2f", old)
    # same bf16 products, f32 sums in another order: at most one bf16 ulp apart
    assert relerr(y.float(), y0.float().double()) < 1e-2
    assert float((y.float() - y0.float()).abs().mean()) < 2e-3


@pytest.mark.parametrize("pad_mode", [1, 0], ids=["reflect", "zeros"])
@pytest.mark.parametrize("shape", [(20, 16, 32), (5, 32, 16), (10, 24, 128), (6, 8, 64)])
def test_first_layer_forward_kernel(shape, pad_mode):
    """conv_c1_fwd.hip (one raw input channel -> 8, bf16): against the f64 reference, against the general kernel, statistics of
    the output from the same launch, masked samples, plain and grouped (four weight sets over one stacked batch)"""
    from passion_amd import ops
    dev = _dev()
    D, H, W = shape
    B = 4
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(B, D, H, W, 1, generator=g) * 1.3 + 0.2).to(torch.bfloat16)
    ws = [(torch.randn(8, 1, 3, 3, 3, generator=g) / 4) for _ in range(4)]
    bs = [torch.randn(8, generator=g) for _ in range(4)]
    mask = torch.tensor([[True], [False], [True], [True]])
    xd, md = x.to(dev), mask.to(dev)

    def ref(wi, bi, samples):
        xm = (x[samples].double() * mask[samples].view(-1, 1, 1, 1, 1)).permute(0, 4, 1, 2, 3)
        xm = F.pad(xm, (1,) * 6, mode="reflect" if pad_mode == 1 else "constant")
        return F.conv3d(xm, wi.to(torch.bfloat16).double(), bi.double())

    ops.prof_read()
    ops.prof_enable(1)
    try:
        wd, bd = ws[0].to(dev).requires_grad_(True), bs[0].to(dev).requires_grad_(True)
        y, st = ops.conv_block([ops.SegIn(xd, mask=md, mask_col=0)], wd, bd, D, H, W, ksize=3, stride=1, pad_mode=pad_mode, want_stats=True)
        wg = [w.to(dev).requires_grad_(True) for w in ws]
        bg = [b.to(dev).requires_grad_(True) for b in bs]
        yg, stg = ops.conv_block_grouped([ops.SegIn(xd, mask=md, mask_col=0)], wg, bg, D, H, W, ksize=3, stride=1, pad_mode=pad_mode,
                                         want_stats=True)
        torch.cuda.synchronize()
    finally:
        ops.prof_enable(0)
    names = [r[4] for r in ops.prof_read()]
    mfma = pad_mode == 1 and W in (32, 64, 128) and H % 8 == 0 and D % 2 == 0 and D >= 4
    vec = H % 16 == 0 and W % 16 == 0                      # the vector kernel's tiles (conv_c1_fwd.hip); everything else: the general kernel
    want = f"conv_c1m_fwd_kernel<{W}>" if mfma else ("conv_c1_fwd_kernel" if vec else "conv_bf16_kernel<3, 1, 1, 0, 8>")
    assert names.count(want) == 2, names
    assert relerr(nc(y.detach().float()), ref(ws[0], bs[0], slice(0, B))) < 1e-2
    for gi in range(4):
        assert relerr(nc(yg[gi:gi + 1].detach().float()), ref(ws[gi], bs[gi], slice(gi, gi + 1))) < 1e-2
    for yy, ss in ((y, st), (yg, stg)):
        want = ops.instnorm_stats(yy.detach())
        std = 1.0 / want[..., 1]
        assert float(((ss[..., 0] - want[..., 0]).abs() / (std + want[..., 0].abs())).max()) < 2e-5
        assert float(((ss[..., 1] - want[..., 1]).abs() / want[..., 1]).max()) < 1e-4
    old = ops.set_option("conv_c1", 0)
    try:
        y0 = ops.conv_block([ops.SegIn(xd, mask=md, mask_col=0)], wd, bd, D, H, W, ksize=3, stride=1, pad_mode=pad_mode)
    finally:
        ops.set_option("conv_c1", old)
    # same bf16 products, f32 sums in another order: at most one bf16 ulp apart
    assert relerr(y.detach().float(), y0.detach().float().double()) < 1e-2
    assert float((y.detach().float() - y0.detach().float()).abs().mean()) < 1e-3
    y.float().sum().backward()
    assert torch.isfinite(wd.grad).all()
