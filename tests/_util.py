"""helpers shared by the parity tests (fixture loading, synthetic inputs, model construction)"""
import os

import numpy as np
import torch
import torch.nn.functional as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_DIMS = dict(basic_dims=4, transformer_basic_dims=64, mlp_dim=128)
FULL_DIMS = dict(basic_dims=8, transformer_basic_dims=512, mlp_dim=4096)


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def synth(B, S, seed=7, blobs=False):
    """same generator as tools/make_golden.py (SURVEY 8d synthetic inputs)"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, S, S, S, generator=g)
    if blobs:
        zz, yy, xx = torch.meshgrid(*[torch.arange(S, dtype=torch.float32)] * 3, indexing="ij")
        r = ((zz - S / 2) ** 2 + (yy - S / 2) ** 2 + (xx - S / 2) ** 2).sqrt()
        lab = torch.zeros(B, S, S, S, dtype=torch.long)
        lab[:, r < S * 0.35] = 2
        lab[:, r < S * 0.25] = 1
        lab[:, r < S * 0.12] = 3
    else:
        lab = torch.randint(0, 4, (B, S, S, S), generator=g)
    return x, F.one_hot(lab, 4).permute(0, 4, 1, 2, 3).double()


def small_weights():
    w = gold("weights_small.npz")
    return {k: torch.from_numpy(w[k].astype(np.float32)) for k in w.files}


def build_product_model(S, small, seed=1037):
    """construct the product Model exactly as tools/make_golden.py constructed the reference one"""
    from passion_amd.models import mmformer
    dims = SMALL_DIMS if small else FULL_DIMS
    mmformer.patch_size = S // 16
    for k, v in dims.items():
        setattr(mmformer, k, v)
    torch.manual_seed(seed)
    m = mmformer.Model(num_cls=4)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n in ("flair_pos", "t1ce_pos", "t1_pos", "t2_pos"):
            getattr(m, n).copy_(torch.randn(getattr(m, n).shape, generator=g) * 0.1)
    if small:
        missing = m.load_state_dict(small_weights(), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    m.eval()
    for k, v in FULL_DIMS.items():
        setattr(mmformer, k, v)
    mmformer.patch_size = 5
    return m


def grad_norms(named_grads):
    out = {}
    for name, g in named_grads:
        if g is None:
            continue
        top = name.split(".")[0]
        out[top] = out.get(top, 0.0) + float(g.double().pow(2).sum())
    return {k: v ** 0.5 for k, v in out.items()}


def rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double().cpu() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double().cpu() if not torch.is_tensor(b) else b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def flash_keep_mask(B, N, p, seed, H=8):
    """numpy restatement of drop_hash (flash_attn.hip): keep[b, h, q, key]"""
    thr = int(p * 65536.0 + 0.5)
    s0, s1 = np.uint64(seed & 0xffffffff), np.uint64(seed >> 32)
    m32, m24 = np.uint64(0xffffffff), np.uint64(0xffffff)
    npair = (N + 1) // 2
    rows = np.arange(B * H * N, dtype=np.uint64)[:, None] * np.uint64(npair)
    pair = ((rows + np.arange(npair, dtype=np.uint64)[None, :]) & m32)
    x = (pair ^ s0) & m32
    x = ((x & m24) * np.uint64(0xB5297A) + (x >> np.uint64(9))) & m32
    x ^= x >> np.uint64(15)
    x = (x + s1) & m32
    x = ((x & m24) * np.uint64(0x68E31D) + (x >> np.uint64(11))) & m32
    x ^= x >> np.uint64(14)
    u = np.stack([x & np.uint64(0xffff), x >> np.uint64(16)], -1).reshape(B * H * N, 2 * npair)[:, :N]
    keep = (u >= thr).reshape(B, H, N, N)
    return torch.from_numpy(keep), 1.0 / (1.0 - thr / 65536.0)
