"""Fused attention (csrc/flash_attn.hip) against a float64 restatement of SelfAttention's core
(reference code/models/mmformer.py:220-224: softmax(q k^T * scale) -> attn_drop -> @ v) on the bf16-rounded inputs, forward and
backward; ragged token counts (the reference's default crop gives N = 125 / 500), the materialising path as a second opinion,
attention dropout with the mask restated in numpy, and run-to-run bit equality (no atomics anywhere in the backward)."""
import numpy as np
import pytest
import torch

from _util import flash_keep_mask as _hash_keep

pytestmark = pytest.mark.gpu
H, DH = 8, 64


def _reference(qkv, go, keep=None, inv_keep=1.0):
    B, N, _ = qkv.shape
    qr = qkv.double().clone().requires_grad_(True)
    q, k, v = qr.reshape(B, N, 3, H, DH).permute(2, 0, 3, 1, 4)
    att = ((q @ k.transpose(-2, -1)) * DH ** -0.5).softmax(-1)
    if keep is not None:
        att = att * keep.double() * inv_keep
    out = (att @ v).transpose(1, 2).reshape(B, N, H * DH)
    out.backward(go.double())
    return out.detach(), qr.grad


def _relerr(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())


def _inputs(B, N, seed=13, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B, N, 3 * H * DH, generator=g) * spread).to(torch.bfloat16)
    go = torch.randn(B, N, H * DH, generator=g).to(torch.bfloat16)
    return qkv, go


@pytest.mark.parametrize("B,N", [(1, 8), (2, 125), (1, 256), (1, 500), (1, 512), (2, 1000), (5, 2048)])
def test_flash_attention_matches_f64(B, N):
    from passion_amd import ops
    dev = torch.device("cuda:0")
    qkv, go = _inputs(B, N)
    outr, gr = _reference(qkv, go)
    qd = qkv.to(dev).requires_grad_(True)
    out = ops.attention(qd, H, DH ** -0.5)
    assert out.grad_fn.name().startswith("FlashAttentionFn")
    out.backward(go.to(dev))
    e_out, e_g = _relerr(out.detach(), outr), _relerr(qd.grad, gr)
    print(f"flash B={B} N={N}: out {e_out:.2e} dqkv {e_g:.2e}")
    assert e_out < 1.5e-2 and e_g < 3e-2
    # q, k, v slots separately (a wrong slot would hide behind the largest one)
    for i, nm in enumerate("qkv"):
        sl = slice(i * H * DH, (i + 1) * H * DH)
        assert _relerr(qd.grad[..., sl], gr[..., sl]) < 3e-2, nm


def test_flash_attention_large_scores_and_lazy_rescale():
    """scores spread over hundreds of log2 units: the lazily moved reference must still give the exact softmax"""
    from passion_amd import ops
    dev = torch.device("cuda:0")
    qkv, go = _inputs(1, 512, seed=5, spread=4.0)
    outr, gr = _reference(qkv, go)
    qd = qkv.to(dev).requires_grad_(True)
    out = ops.attention(qd, H, DH ** -0.5)
    out.backward(go.to(dev))
    assert _relerr(out.detach(), outr) < 1.5e-2 and _relerr(qd.grad, gr) < 4e-2


def test_flash_agrees_with_the_materialising_path():
    from passion_amd import ops
    dev = torch.device("cuda:0")
    qkv, go = _inputs(2, 512, seed=3)
    res = []
    for flash in (True, False):
        ops.FLASH = flash
        try:
            qd = qkv.to(dev).requires_grad_(True)
            out = ops.attention(qd, H, DH ** -0.5)
            out.backward(go.to(dev))
            res.append((out.detach().float(), qd.grad.float()))
        finally:
            ops.FLASH = True
    assert float((res[0][0] - res[1][0]).abs().max() / res[1][0].abs().max()) < 2e-2
    assert float((res[0][1] - res[1][1]).abs().max() / res[1][1].abs().max()) < 4e-2


@pytest.mark.parametrize("B,N,p", [(1, 125, 0.5), (2, 512, 0.1), (5, 2048, 0.1)])
def test_flash_attention_dropout_mask_forward_and_backward(B, N, p):
    """the kernel's keep bits are the numpy-restated hash's, in the forward and (regenerated) in both backward kernels"""
    from passion_amd import ops
    dev = torch.device("cuda:0")
    qkv, go = _inputs(B, N, seed=7)
    torch.manual_seed(99)
    seed = ops._draw_seed()
    keep, inv_keep = _hash_keep(B, N, p, seed)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
    outr, gr = _reference(qkv, go, keep, inv_keep)
    qd = qkv.to(dev).requires_grad_(True)
    torch.manual_seed(99)
    out = ops.attention(qd, H, DH ** -0.5, p)
    out.backward(go.to(dev))
    e_out, e_g = _relerr(out.detach(), outr), _relerr(qd.grad, gr)
    print(f"flash dropout B={B} N={N} p={p}: out {e_out:.2e} dqkv {e_g:.2e}")
    assert e_out < 1.5e-2 and e_g < 3e-2


@pytest.mark.parametrize("B,N,split,p", [(1, 2048, 0, 0.0), (1, 2048, 4, 0.1), (1, 500, 8, 0.1), (2, 125, 2, 0.0), (4, 512, 0, 0.1), (1, 100, 3, 0.1), (5, 2048, 2, 0.1)])
def test_flash_attention_split_keys(B, N, split, p):
    """small batches split the key (query) range of a tile over several workgroups and merge f32 partials (option "flash_split",
    0 = the automatic choice of the benchmarked step: B = 1, N = 2048 -> 4): same mask, same result as the f64 restatement,
    and within bf16 rounding of the unsplit kernels"""
    from passion_amd import ops
    dev = torch.device("cuda:0")
    qkv, go = _inputs(B, N, seed=31)
    keep, inv_keep = None, 1.0
    if p:
        torch.manual_seed(99)
        keep, inv_keep = _hash_keep(B, N, p, ops._draw_seed())
    outr, gr = _reference(qkv, go, keep, inv_keep)
    res = {}
    for sp in (1, split):
        old = ops.set_option("flash_split", sp)
        try:
            qd = qkv.to(dev).requires_grad_(True)
            torch.manual_seed(99)
            out = ops.attention(qd, H, DH ** -0.5, p)
            out.backward(go.to(dev))
            res[sp] = (out.detach().clone(), qd.grad.clone())
        finally:
            ops.set_option("flash_split", old)
    e_out, e_g = _relerr(res[split][0], outr), _relerr(res[split][1], gr)
    d_out, d_g = _relerr(res[split][0], res[1][0].double().cpu()), _relerr(res[split][1], res[1][1].double().cpu())
    print(f"flash split={split} B={B} N={N} p={p}: vs f64 out {e_out:.2e} dqkv {e_g:.2e}; vs unsplit out {d_out:.2e} dqkv {d_g:.2e}")
    assert e_out < 1.5e-2 and e_g < 3e-2 and d_out < 1e-2 and d_g < 1.6e-2
    for i, nm in enumerate("qkv"):
        sl = slice(i * H * DH, (i + 1) * H * DH)
        assert _relerr(res[split][1][..., sl], gr[..., sl]) < 3e-2, nm


@pytest.mark.parametrize("B", [5, 1])
def test_flash_attention_is_bitwise_reproducible(B):
    from passion_amd import ops
    dev = torch.device("cuda:0")
    qkv, go = _inputs(B, 2048, seed=21)
    outs = []
    for _ in range(2):
        qd = qkv.to(dev).requires_grad_(True)
        torch.manual_seed(4)
        out = ops.attention(qd, H, DH ** -0.5, 0.1)
        out.backward(go.to(dev))
        outs.append((out.detach().clone(), qd.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
