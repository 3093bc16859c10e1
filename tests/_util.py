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


def synth(B, S, seed=7, blobs=False, radii=(0.35, 0.25, 0.12)):
    """same generator as tools/make_golden.py (SURVEY 8d synthetic inputs)"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, S, S, S, generator=g)
    if blobs:
        zz, yy, xx = torch.meshgrid(*[torch.arange(S, dtype=torch.float32)] * 3, indexing="ij")
        r = ((zz - S / 2) ** 2 + (yy - S / 2) ** 2 + (xx - S / 2) ** 2).sqrt()
        lab = torch.zeros(B, S, S, S, dtype=torch.long)
        lab[:, r < S * radii[0]] = 2
        lab[:, r < S * radii[1]] = 1
        lab[:, r < S * radii[2]] = 3
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


def _drop_hash(x, s0, s1):
    """two rounds of (24-bit multiply-add, xorshift) on uint64-held 32-bit words: drop_hash of flash_attn.hip"""
    m32, m24 = np.uint64(0xffffffff), np.uint64(0xffffff)
    x = (x ^ s0) & m32
    x = ((x & m24) * np.uint64(0xB5297A) + (x >> np.uint64(9))) & m32
    x ^= x >> np.uint64(15)
    x = (x + s1) & m32
    x = ((x & m24) * np.uint64(0x68E31D) + (x >> np.uint64(11))) & m32
    x ^= x >> np.uint64(14)
    return x


def flash_keep_mask(B, N, p, seed, H=8):
    """numpy restatement of the attention-dropout mask of flash_attn.hip: keep[b, h, q, key].  One 32-bit word per (query row, key
    pair): word = mix(R(row) + C(pair)), R / C two-round hashes of (seed, (b H + h) N + q) / (seed, pair), mix = xorshift + one
    24-bit multiply-add; key 2 kp takes the low 16-bit field, key 2 kp + 1 the high one; keep = signed field >= thr - 32768"""
    thr = int(p * 65536.0 + 0.5)
    s0, s1 = np.uint64(seed & 0xffffffff), np.uint64(seed >> 32)
    m32, m24 = np.uint64(0xffffffff), np.uint64(0xffffff)
    npair = (N + 1) // 2
    R = _drop_hash(np.arange(B * H * N, dtype=np.uint64), s0, s1)
    C = _drop_hash(np.arange(npair, dtype=np.uint64) ^ np.uint64(0x5BD1E995), s1, (s0 + np.uint64(0x2545F491)) & m32)
    x = (R[:, None] + C[None, :]) & m32
    x ^= x >> np.uint64(15)
    x = ((x & m24) * np.uint64(0x68E31D) + (x >> np.uint64(11))) & m32
    u = np.stack([x & np.uint64(0xffff), x >> np.uint64(16)], -1).reshape(B * H * N, 2 * npair)[:, :N]
    keep = (((u + np.uint64(32768)) & np.uint64(0xffff)) >= thr).reshape(B, H, N, N)       # signed 16-bit field >= thr - 32768
    return torch.from_numpy(keep), 1.0 / (1.0 - thr / 65536.0)


# ---- multi-step fixtures (tools/make_golden_traj.py) ------------------------------------------------------------------------
def signal_volume(S, means, seed=7, radii=None):
    """blob labels and x = 0.5 * noise + a class-dependent mean per modality (make_golden_traj.signal_volume); radii: the three sphere
    radii of the labels as fractions of S (a fixture that stores them; default: make_golden.synth's)"""
    x, target = synth(1, S, seed=seed, blobs=True, **({} if radii is None else {"radii": tuple(float(v) for v in radii)}))
    lab = target.argmax(1)
    x = 0.5 * x + torch.as_tensor(np.asarray(means), dtype=torch.float32)[:, lab[0]].unsqueeze(0)
    return x, target


def unpack2(packed, n):
    p = np.asarray(packed)
    out = np.empty((p.size, 4), dtype=np.uint8)
    for j in range(4):
        out[:, j] = (p >> (2 * j)) & 3
    return out.reshape(-1)[:n]


def replay_trajectory(g, dtype, dev, on_step=None, force_gates=None):
    """the product takes the fixture's K AdamW(amsgrad) steps from the same seed / inputs / learning rate (dropout off, as the
    fixture) -> (model, per-step losses [K + 1], final prediction (1,4,S,S,S)); the step is passion_amd.train's.
    force_gates [K + 1][4]: PASSION's relative-preference gate of every step is the given one (the reference's) instead of the sign of
    the product's own relative distances"""
    from passion_amd import train as ptrain
    S, K, passion = int(g["S"]), int(g["K"]), bool(g["passion"])
    m = build_product_model(S, small=False).to(dev)
    m.mask_type, m.use_passion, m.is_training, m.compute_dtype = "idt", passion, True, dtype
    x, target = signal_volume(S, g["means"], radii=g["radii"] if "radii" in g.files else None)
    x, target, mask = x.to(dev), target.to(dev), torch.from_numpy(g["mask"]).to(dev)
    opt = ptrain.make_optimizer(m, lr=float(g["lr"]), weight_decay=1e-4)
    losses, pred = [], None
    for it in range(K + 1):
        outs = m(x, mask, target=target, temp=4.0)
        forced = torch.from_numpy(np.asarray(force_gates[it], dtype=np.float32)) if force_gates is not None else None
        loss = ptrain.assemble_loss(outs, target, mask, "idt", passion, rp_override=forced)
        losses.append(loss.detach())
        pred = outs[0].detach()
        if on_step is not None:
            on_step(it, outs, loss)
        if it == K:
            break
        opt.zero_grad()
        loss.backward()
        opt.step()
    return m, torch.stack(losses).double().cpu().numpy(), pred


def mask_dice(a, b):
    """mean Dice (WT, TC, ET as predict.py:78-124 forms them) between two integer label volumes + the differing-voxel count"""
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    d = []
    for cls in ((1, 2, 3), (1, 3), (3,)):
        oa, ob = np.isin(a, cls), np.isin(b, cls)
        d.append((2.0 * (oa & ob).sum() + 1e-8) / (oa.sum() + ob.sum() + 1e-8))
    mask_dice.last = [float(v) for v in d]            # (WT, TC, ET) of the call just made
    return float(np.mean(d)), int((a != b).sum())


def build_model_grid(grid, small=True, seed=11):
    """product Model for a (pD, pH, pW) token grid (a (16 pD, 16 pH, 16 pW) volume) -- beyond the reference, whose grid is a cube
    (mmformer.py:21,342-345): conv weights of the small-constants fixture where shapes agree (small=True), random position
    embeddings.  -> model in eval() mode on the CPU"""
    from passion_amd.models import mmformer
    keep = {k: getattr(mmformer, k) for k in FULL_DIMS}
    keep_ps = mmformer.patch_size
    try:
        for k, v in (SMALL_DIMS if small else FULL_DIMS).items():
            setattr(mmformer, k, v)
        mmformer.patch_size = tuple(grid)
        torch.manual_seed(seed)
        m = mmformer.Model(num_cls=4)
    finally:
        for k, v in keep.items():
            setattr(mmformer, k, v)
        mmformer.patch_size = keep_ps
    sd = m.state_dict()
    with torch.no_grad():
        if small:
            for k, v in small_weights().items():
                if sd[k].shape == v.shape:
                    sd[k].copy_(v)
        g = torch.Generator().manual_seed(seed + 1)
        for n in ("flair_pos", "t1ce_pos", "t1_pos", "t2_pos"):
            sd[n].copy_(torch.randn(sd[n].shape, generator=g) * 0.1)
    m.eval()
    return m


def synth_box(B, shape, seed=7):
    """synthetic (B,4,D,H,W) volume + ellipsoid blob labels for a non-cube shape"""
    g = torch.Generator().manual_seed(seed)
    D, H, W = shape
    x = torch.randn(B, 4, D, H, W, generator=g)
    zz, yy, xx = torch.meshgrid(torch.arange(D, dtype=torch.float32), torch.arange(H, dtype=torch.float32),
                                torch.arange(W, dtype=torch.float32), indexing="ij")
    r = (((zz - D / 2) / D) ** 2 + ((yy - H / 2) / H) ** 2 + ((xx - W / 2) / W) ** 2).sqrt()
    lab = torch.zeros(B, D, H, W, dtype=torch.long)
    lab[:, r < 0.35] = 2
    lab[:, r < 0.25] = 1
    lab[:, r < 0.12] = 3
    return x, F.one_hot(lab, 4).permute(0, 4, 1, 2, 3).double()


M2F_SMALL = dict(basic_dims=4, mlp_dim=128)
M2F_FULL = dict(basic_dims=8, mlp_dim=4096)


def build_m2ftrans(S, small=True, seed=1037):
    """product m2ftrans.Model built like tools/make_golden_m2f.py builds the reference one (seed 1037, pos from seed 1038);
    small=True: the small-constants layout + the fixture's weights (strict load)"""
    from passion_amd.models import m2ftrans
    keep = {k: getattr(m2ftrans, k) for k in M2F_FULL}
    keep_ps = m2ftrans.patch_size
    try:
        for k, v in (M2F_SMALL if small else M2F_FULL).items():
            setattr(m2ftrans, k, v)
        m2ftrans.patch_size = S // 16
        torch.manual_seed(seed)
        m = m2ftrans.Model(num_cls=4)
    finally:
        for k, v in keep.items():
            setattr(m2ftrans, k, v)
        m2ftrans.patch_size = keep_ps
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        m.pos.copy_(torch.randn(m.pos.shape, generator=g) * 0.1)
    m.eval()
    return m
