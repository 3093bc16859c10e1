"""ops.token_drop (csrc/token_ops.hip token_drop_*): the elementwise glue of the reference's Transformer blocks in one launch each way --
x + dropout(dropout(proj)) (code/models/mmformer.py:226,247,256), dropout(gelu(h)) (:271-273), x + dropout(h) (:275,247) -- against the
torch ops it replaces: exactly without dropout; with dropout against the same expression under the kernel's own keep mask (restated in
numpy from the seed), forward and backward; keep rates of one and of two dropouts in a row; fresh masks per call, repeatable from a seed."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _keep_mask(n, p, seed, ndrop):
    """numpy restatement of tok_keep: two-round (24-bit multiply-add, xorshift) hash of (seed, element index); the low 16-bit field is
    the first dropout's uniform, the high one the second's; keep = field >= round(p * 65536)"""
    thr = int(p * 65536.0 + 0.5)
    m32, m24 = np.uint64(0xffffffff), np.uint64(0xffffff)
    s0, s1 = np.uint64(seed & 0xffffffff), np.uint64(seed >> 32)
    x = (np.arange(n, dtype=np.uint64) ^ s0) & m32
    x = ((x & m24) * np.uint64(0xB5297A) + (x >> np.uint64(9))) & m32
    x ^= x >> np.uint64(15)
    x = (x + s1) & m32
    x = ((x & m24) * np.uint64(0x68E31D) + (x >> np.uint64(11))) & m32
    x ^= x >> np.uint64(14)
    k = (x & np.uint64(0xffff)) >= thr
    if ndrop == 2:
        k &= (x >> np.uint64(16)) >= thr
    keep = 1.0 - thr / 65536.0
    return torch.from_numpy(k), 1.0 / (keep ** ndrop)


@pytest.mark.parametrize("ydt", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("mode", ["drop2_add", "gelu_drop", "drop_add"])
def test_token_drop_matches_the_torch_ops(mode, ydt):
    from passion_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    R, C = 96, 512 if mode != "gelu_drop" else 4096
    y = torch.randn(R, C, generator=g).to(ydt).to(dev)
    x = torch.randn(R, C, generator=g).to(dev) if mode != "gelu_drop" else None
    ndrop, act = (2, 0) if mode == "drop2_add" else ((1, 1) if mode == "gelu_drop" else (1, 0))
    go = torch.randn(R, C, generator=g).to(torch.float32 if x is not None else ydt).to(dev)

    def ref(yv, xv, mask, scale):
        z = F.gelu(yv) if act else yv
        if mask is not None:
            z = (z.float() * (mask.to(dev).view(R, C).float() * scale)).to(z.dtype)
        return z + xv if xv is not None else z

    # ---- no dropout: the torch expression itself, bit for bit (forward) / to rounding (GELU's derivative in the backward)
    y1 = y.clone().requires_grad_(True)
    x1 = x.clone().requires_grad_(True) if x is not None else None
    out = ops.token_drop(y1, x1, 0.0, ndrop, act)
    out.backward(go)
    y2 = y.clone().requires_grad_(True)
    x2 = x.clone().requires_grad_(True) if x is not None else None
    r = ref(y2, x2, None, 1.0)
    r.backward(go)
    assert out.dtype == r.dtype and torch.equal(out, r)
    assert float((y1.grad.float() - y2.grad.float()).abs().max()) <= (2e-2 if ydt == torch.bfloat16 else 1e-5) * float(y2.grad.float().abs().max())
    if x is not None:
        assert torch.equal(x1.grad, x2.grad)
    # ---- dropout 0.1: the same expression under the kernel's own mask
    torch.manual_seed(77)
    seed = ops._draw_seed()
    mask, scale = _keep_mask(R * C, 0.1, seed, ndrop)
    assert abs(float(mask.float().mean()) - 0.9 ** ndrop) < 5e-3
    y1 = y.clone().requires_grad_(True)
    x1 = x.clone().requires_grad_(True) if x is not None else None
    torch.manual_seed(77)
    out = ops.token_drop(y1, x1, 0.1, ndrop, act)
    out.backward(go)
    y2 = y.clone().double().requires_grad_(True)
    x2 = x.clone().double().requires_grad_(True) if x is not None else None
    z = F.gelu(y2) if act else y2
    r = z * (mask.to(dev).view(R, C).double() * scale)
    if x2 is not None:
        r = r + x2
    r.backward(go.double())
    tol = 1e-2 if (ydt == torch.bfloat16 and x is None) else 1e-5
    assert float((out.double() - r).abs().max()) <= tol * float(r.abs().max())
    dropped = ~mask.to(dev).view(R, C)
    assert float(y1.grad[dropped].abs().max()) == 0.0                      # a dropped element passes no gradient
    assert float((y1.grad.double() - y2.grad).abs().max()) <= (1e-2 if ydt == torch.bfloat16 else 1e-5) * float(y2.grad.abs().max())
    if x is not None:
        assert torch.equal(x1.grad, go)
    # ---- a new call draws a new mask; the same seed repeats it
    out2 = ops.token_drop(y, x, 0.1, ndrop, act)
    assert not torch.equal(out2, out.detach())
    torch.manual_seed(77)
    assert torch.equal(ops.token_drop(y, x, 0.1, ndrop, act), out.detach())


def test_transformer_block_with_fused_glue_equals_the_module_chain_without_dropout():
    """Residual(PreNormDrop(SelfAttention)) / Residual(PreNorm(FeedForward)) in training mode with p = 0: the fused path gives the
    unfused modules' values bit for bit (bf16 autocast), so every dropout-free parity / trajectory test of the suite covers it"""
    from passion_amd.models import mmformer
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    tr = mmformer.Transformer(embedding_dim=512, depth=1, heads=8, mlp_dim=4096).to(dev)
    for m in tr.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    tr.train()
    x = torch.randn(1, 512, 512, device=dev)
    pos = torch.randn(1, 512, 512, device=dev) * 0.02
    outs = []
    for fuse in (True, False):
        mmformer.FUSE_TOKEN_GLUE = fuse
        try:
            xi = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                o = tr(xi, pos)
            o.float().pow(2).sum().backward()
            outs.append((o.detach().clone(), xi.grad.clone(), tr.cross_ffn_list[0].fn.fn.net[0].weight.grad.clone()))
            tr.zero_grad()
        finally:
            mmformer.FUSE_TOKEN_GLUE = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 2e-2 * float(outs[1][1].abs().max())
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= 2e-2 * float(outs[1][2].abs().max())
