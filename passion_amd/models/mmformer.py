"""mmformer backbone + PASSION self-distillation losses on the MI355X HIP kernels.

Drop-in for the reference's code/models/mmformer.py: `Model(num_cls=4)` with the plain attributes
`mask_type`, `use_passion`, `is_training` set after construction (train.py:91-92,212) and
`forward(x, mask, target=None, temp=1.0)` returning softmax(fuse) | (pred, prm_loss, sep_loss) |
(pred, prm_loss, sep_loss, kl_loss, proto_loss, dist) exactly as mmformer.py:635,657,659.  The
module tree, parameter names, shapes, registration order and initialisation (hence the RNG stream
of torch.manual_seed) are the reference's, so its checkpoints load with strict=True.

What differs is everything underneath: activations are channels-last (B,D,H,W,C) in f32 or bf16
(`compute_dtype`), every conv / InstanceNorm / LeakyReLU / upsample / softmax / loss / attention
core is a HIP kernel from libpassion_hip.so (passion_amd.ops), MaskModal + stack + cat never
materialise (they are per-segment masks of the conv loader), InstanceNorm statistics of a tensor are
computed once and shared by all its consumers (the five decoder_fuse passes re-use them).

Module-level constants are read at construction time, as in the reference, so
`mmformer.patch_size = S // 16` selects a crop edge S (the reference hard-codes 80 -> 5).  Beyond the reference:
`mmformer.patch_size = (pD, pH, pW)` builds the position embeddings for a NON-CUBE token grid (a (16 pD, 16 pH, 16 pW) volume:
BASELINE.json configs[4]'s 240 x 240 x 160 is (10, 15, 15) with D first); an int keeps the reference's cube.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from ..utils import criterions as crit
from .blocks import Feat, _ln, _ln_ok, fusion_prenorm, general_conv3d_prenorm, run_conv, run_pointwise

basic_dims = 8
transformer_basic_dims = 512
mlp_dim = 4096
num_heads = 8
depth = 1
num_cls = 4
num_modals = 4
patch_size = 5
H = W = Z = 80

MODALS = ("flair", "t1ce", "t1", "t2")
import os as _os
ENC_STREAMS = _os.environ.get("PASSION_ENC_STREAMS", "1") != "0"
# Parameters that are first touched on a side stream (encoders, intra-modal transformers) get their AccumulateGrad node on
# that stream while e.g. the position embeddings also receive a gradient from the main stream: autograd synchronises the
# two correctly and says so in a warning; the mismatch is intended.
if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
INTRA_STREAMS = _os.environ.get("PASSION_INTRA_STREAMS", "1") != "0"   # intra-modal transformers stay on their encoder's stream


def n_tokens(ps):
    """tokens per modality of a token grid given as the reference's cube edge (int) or as (pD, pH, pW)"""
    return int(ps) ** 3 if isinstance(ps, int) else int(np.prod(ps))


def _up2(f):
    """x2 trilinear; the result goes into a pre-norm conv, so the statistics come along (one pass, ops.upsample_with_stats)"""
    from .blocks import FUSE_STATS
    if FUSE_STATS and f.t.is_cuda:
        y, st = ops.upsample_with_stats(f.t, 2)
        return Feat(y, f.mask, f.mask_col, _stats=[st])
    return Feat(ops.upsample(f.t, 2), f.mask, f.mask_col)


class Encoder(nn.Module):
    """five-level pre-norm residual encoder of one modality (reference mmformer.py:24-64)"""

    def __init__(self):
        super().__init__()
        c = basic_dims
        self.e1_c1 = nn.Conv3d(in_channels=1, out_channels=c, kernel_size=3, stride=1, padding=1,
                               padding_mode="reflect", bias=True)
        self.e1_c2 = general_conv3d_prenorm(c, c, pad_type="reflect")
        self.e1_c3 = general_conv3d_prenorm(c, c, pad_type="reflect")
        for lvl in range(2, 6):
            ci, co = c * 2 ** (lvl - 2), c * 2 ** (lvl - 1)
            setattr(self, f"e{lvl}_c1", general_conv3d_prenorm(ci, co, stride=2, pad_type="reflect"))
            setattr(self, f"e{lvl}_c2", general_conv3d_prenorm(co, co, pad_type="reflect"))
            setattr(self, f"e{lvl}_c3", general_conv3d_prenorm(co, co, pad_type="reflect"))

    def forward(self, x, modal, mask, dtype, outs=None, stats=None):
        """x: the whole (B,4,S,S,S) input; this encoder reads modality `modal` as a C=1 segment.
        outs / stats: optional per-level output slots (this modality's slice of the buffers that stack the
        four modalities for decoder_sep) and the matching InstanceNorm-statistics slots."""
        B, K, D, Hh, Ww = x.shape
        vol = D * Hh * Ww
        seg = ops.SegIn(x, mask=mask, mask_col=modal, C=1, offset=modal * vol, bstride=K * vol)
        cache = self.e1_c1.__dict__.setdefault("_passion_pack", {})
        from .blocks import FUSE_STATS
        if FUSE_STATS:
            y0, st0 = ops.conv_block([seg], self.e1_c1.weight, self.e1_c1.bias, D, Hh, Ww, ksize=3, stride=1,
                                     pad_mode=L.PAD_REFLECT, out_dtype=dtype, pack_cache=cache, want_stats=True)
            f = Feat(y0, _stats=[st0])
        else:
            f = Feat(ops.conv_block([seg], self.e1_c1.weight, self.e1_c1.bias, D, Hh, Ww, ksize=3, stride=1,
                                    pad_mode=L.PAD_REFLECT, out_dtype=dtype, pack_cache=cache))
        feats = []
        for lvl in range(1, 6):
            if lvl > 1:
                f = getattr(self, f"e{lvl}_c1")(f)
            f = getattr(self, f"e{lvl}_c3")(getattr(self, f"e{lvl}_c2")(f), residual=f.t,
                                            out=outs[lvl - 1] if outs is not None else None,
                                            stats_out=stats[lvl - 1] if stats is not None else None,
                                            want_stats=lvl < 5)      # level 5 feeds the token conv / an upsampling only
            feats.append(f)
        return feats


GROUPED_ENC = _os.environ.get("PASSION_GROUPED_ENC", "1") != "0"   # the four modality encoders as one grouped launch per layer


def encoders_grouped(encs, x, mask, dtype):
    """The four modality encoders (reference mmformer.py:385-388, 397-415: four Encoder() instances of identical shape) over the
    stacked batch [flair | t1ce | t1 | t2] (4B samples): every layer is ONE grouped launch (ops.conv_block_grouped: sample s
    uses the weights of encoder s // B) instead of four launches on four streams -- a quarter of the launches, and deep-level
    kernels four times the size.  x: (B,4,D,H,W); mask: (B,4) bool or None (idt: absent modalities read as zeros).
    -> five Feats of shape (4B, D_l, H_l, W_l, C_l) with their statistics."""
    B, K, D, H, W = x.shape
    xs = torch.empty(K, B, D, H, W, 1, dtype=dtype, device=x.device)
    xs.view(K, B, D, H, W).copy_(x.transpose(0, 1))             # cast + modality-major restack in one pass
    xs = xs.view(K * B, D, H, W, 1)
    gmask = mask.t().reshape(K * B, 1).contiguous() if mask is not None else None      # row m * B + b = mask[b, m]

    def gconv(name, f, residual=None, want_stats=True, first=False, pass_x=False):
        convs = [getattr(e, name) if first else getattr(e, name).conv for e in encs]
        c0 = convs[0]
        d, h, w = f.t.shape[1:4]
        seg = ops.SegIn(f.t, mask=gmask, mask_col=0) if first else f.seg(True)
        r = ops.conv_block_grouped([seg], [c.weight for c in convs], [c.bias for c in convs], d, h, w, ksize=c0.kernel_size[0],
                                   stride=c0.stride[0], pad_mode=L.PAD_REFLECT, residual=residual, out_dtype=dtype,
                                   want_stats=want_stats, pass_x=pass_x)
        if pass_x:
            return (Feat(r[0], _stats=[r[1]]),) + tuple(r[2:])
        return Feat(r[0], _stats=[r[1]]) if want_stats else Feat(r)

    f = gconv("e1_c1", Feat(xs), first=True)
    feats = []
    for lvl in range(1, 6):
        if lvl > 1 and RES_FUSE:
            # the level's feature goes to decoder_sep, to the fusion blocks AND into the next level's stride-2 conv: that conv hands it on
            # (twice: decoder_sep and the fusion path are different branches of the backward), the others read the handed-on tensors, and
            # their gradients join the conv's own input gradient inside its backward (passion_seg_t.dx_add, dx_add2)
            fprev = f
            if RES_FUSE >= 2:
                f, xa, xb = gconv(f"e{lvl}_c1", f, pass_x=2)
            else:
                f, xa = gconv(f"e{lvl}_c1", f, pass_x=1)
                xb = None
            feats[-1] = Feat(xa, fprev.mask, fprev.mask_col, fprev._stats, fprev.passes, fprev.stats_out, fprev._act, fprev.c_real)
            feats[-1].alt = xb
        elif lvl > 1:
            f = gconv(f"e{lvl}_c1", f)
        if RES_FUSE:     # x + c3(c2(x)): c2 hands x on to c3's residual input, the two gradients of x are summed inside c2's backward
            y2, xres = gconv(f"e{lvl}_c2", f, pass_x=True)
            f = gconv(f"e{lvl}_c3", y2, residual=xres, want_stats=lvl < 5)
        else:
            f = gconv(f"e{lvl}_c3", gconv(f"e{lvl}_c2", f), residual=f.t, want_stats=lvl < 5)
        feats.append(f)
    return feats


from .blocks import RES_FUSE      # (gradient sums inside the conv backward: see encoders_grouped)
SEP_LOSS_FIRST = _os.environ.get("PASSION_SEP_LOSS_FIRST", "1") != "0"
TCL_EARLY = _os.environ.get("PASSION_TCL_EARLY", "1") != "0"           # the label conversion at the top of the forward on the loss stream
DS_EARLY = _os.environ.get("PASSION_DS_EARLY", "1") != "0"             # deep-supervision losses as soon as a level's logits exist (needs TCL_EARLY)
LOSS_STREAMS = _os.environ.get("PASSION_LOSS_STREAMS", "1") != "0"     # deep-supervision / decoder_sep losses on side streams
FUSE_STREAM = _os.environ.get("PASSION_FUSE_STREAM", "1") != "0"       # fusion blocks of levels 4..1 on a side stream
GROUPED_INTRA = _os.environ.get("PASSION_GROUPED_INTRA", "1") != "0"   # the four intra-modal transformers as batched GEMMs


def _gstack(params):
    return torch.stack(list(params))


def intra_grouped(model, x5, mask, B, bf16):
    """The four token convs + intra-modal transformers (reference mmformer.py:417-433: *_encode_conv, *_pos, *_transformer, one
    of each per modality, identical shapes) over the stacked level-5 features (4B, p, p, p, C): every GEMM is ONE batched GEMM
    with four weight sets (ops.linear_grouped), LayerNorm / GELU / dropout / residuals run once over the stack, attention is one
    call with batch 4B -- ~30 launches instead of 4 x 25 per direction.  Same operations in the same order and precisions as
    Transformer.forward under bf16 autocast (LayerNorm and the residual stream in f32, GEMM operands in bf16).
    -> (4, B, N, C) tokens, already multiplied by the modality mask."""
    G = 4
    convs = [getattr(model, f"{m}_encode_conv") for m in MODALS]
    trs = [getattr(model, f"{m}_transformer") for m in MODALS]
    assert all(t.depth == 1 for t in trs), "intra_grouped covers the reference's depth-1 transformers"
    N = x5.shape[1] * x5.shape[2] * x5.shape[3]
    C = model.token_dim
    x = x5.view(G, B * N, x5.shape[-1])
    tok = ops.linear_grouped(x, [c.weight for c in convs], [c.bias for c in convs], bf16)            # (G, B*N, C)
    pos = _gstack([getattr(model, f"{m}_pos") for m in MODALS]).view(G, 1, N, C)
    xr = tok.view(G, B, N, C) + pos                                                                  # f32 residual stream
    xr = xr.view(G, B * N, C)
    att = [t.cross_attention_list[0].fn for t in trs]       # PreNormDrop(norm, dropout, SelfAttention)
    ffn = [t.cross_ffn_list[0].fn for t in trs]             # PreNorm(norm, FeedForward)
    training = trs[0].training

    def gnorm(mods, t):
        if _ln_ok(t):                                # one launch, four affine sets, straight into the GEMM's bf16 operand
            return ops.ln_act(t, mods, out_dtype=torch.bfloat16 if bf16 else None)
        n = F.layer_norm(t, (C,), None, None, mods[0].eps)
        return torch.addcmul(_gstack([m.bias for m in mods]).unsqueeze(1), n, _gstack([m.weight for m in mods]).unsqueeze(1))

    sa = [a.fn for a in att]
    h = gnorm([a.norm for a in att], xr)
    qkv = ops.linear_grouped(h, [m.qkv.weight for m in sa], None, bf16)                              # (G, B*N, 3C)
    p_att = sa[0].attn_drop.p if training else 0.0
    o = ops.attention(qkv.view(G * B, N, 3 * C), sa[0].num_heads, sa[0].scale, p_att)               # batch 4B
    o = ops.linear_grouped(o.reshape(G, B * N, C), [m.proj.weight for m in sa], [m.proj.bias for m in sa], bf16)
    fuse = FUSE_TOKEN_GLUE and training and xr.dtype == torch.float32 and sa[0].proj_drop.p == att[0].dropout.p
    if fuse and ops.token_drop_ok(o, xr):
        xr = ops.token_drop(o, xr, att[0].dropout.p, 2, 0)
    else:
        o = F.dropout(o, sa[0].proj_drop.p, training)
        o = F.dropout(o, att[0].dropout.p, training)
        xr = o + xr
    ff = [f.fn.net for f in ffn]
    h = gnorm([f.norm for f in ffn], xr)
    h = ops.linear_grouped(h, [n[0].weight for n in ff], [n[0].bias for n in ff], bf16)
    h = ops.token_drop(h, None, ff[0][2].p, 1, 1) if (fuse and ops.token_drop_ok(h)) else F.dropout(F.gelu(h), ff[0][2].p, training)
    h = ops.linear_grouped(h, [n[3].weight for n in ff], [n[3].bias for n in ff], bf16)
    if fuse and ops.token_drop_ok(h, xr):
        xr = ops.token_drop(h, xr, ff[0][4].p, 1, 0)
    else:
        h = F.dropout(h, ff[0][4].p, training)
        xr = h + xr
    gm = mask.t().reshape(G, B, 1, 1).to(xr.dtype)
    return xr.view(G, B, N, C) * gm


def _decoder_convs(mod):
    c = basic_dims
    for lvl in (4, 3, 2, 1):
        co = c * 2 ** (lvl - 1)
        setattr(mod, f"d{lvl}_c1", general_conv3d_prenorm(co * 2, co, pad_type="reflect"))
        setattr(mod, f"d{lvl}_c2", general_conv3d_prenorm(co * 2, co, pad_type="reflect"))
        setattr(mod, f"d{lvl}_out", general_conv3d_prenorm(co, co, k_size=1, padding=0, pad_type="reflect"))


class Decoder_sep(nn.Module):
    """per-modality U-Net decoder, shared weights (reference mmformer.py:66-114); returns softmax probs
    (channels-last f32)"""

    def __init__(self, num_cls=4):
        super().__init__()
        c = basic_dims
        for lvl in (4, 3, 2, 1):
            co = c * 2 ** (lvl - 1)
            setattr(self, f"d{lvl}", nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True))
            setattr(self, f"d{lvl}_c1", general_conv3d_prenorm(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_c2", general_conv3d_prenorm(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_out", general_conv3d_prenorm(co, co, k_size=1, padding=0, pad_type="reflect"))
        self.seg_layer = nn.Conv3d(in_channels=c, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x1, x2, x3, x4, x5, probs=True):
        skips = (x1, x2, x3, x4)
        h = self.d4_c1(_up2(x5))
        for lvl in (4, 3, 2, 1):
            h = getattr(self, f"d{lvl}_out")(getattr(self, f"d{lvl}_c2")([h, skips[lvl - 1]]), want_stats=False)
            if lvl > 1:
                h = getattr(self, f"d{lvl - 1}_c1")(_up2(h))
        logits = run_conv(self.seg_layer, h, norm=False, out_dtype=torch.float32).t
        return ops.softmax_classes(logits, 1.0, "cl") if probs else logits      # (logits: the loss forms the softmax itself, ops.sep_wce_dice)


class Decoder_fuse(nn.Module):
    """fusion decoder with deep supervision (reference mmformer.py:116-189); returns channels-last
    logits, the four deep-supervision logits and the five feature maps"""

    def __init__(self, num_cls=4):
        super().__init__()
        c = basic_dims
        _decoder_convs(self)
        self.seg_d4 = nn.Conv3d(in_channels=c * 16, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.seg_d3 = nn.Conv3d(in_channels=c * 8, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.seg_d2 = nn.Conv3d(in_channels=c * 4, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.seg_d1 = nn.Conv3d(in_channels=c * 2, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.seg_layer = nn.Conv3d(in_channels=c, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.up2 = nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
        self.RFM5 = fusion_prenorm(in_channel=c * 16, num_cls=num_cls)
        self.RFM4 = fusion_prenorm(in_channel=c * 8, num_cls=num_cls)
        self.RFM3 = fusion_prenorm(in_channel=c * 4, num_cls=num_cls)
        self.RFM2 = fusion_prenorm(in_channel=c * 2, num_cls=num_cls)
        self.RFM1 = fusion_prenorm(in_channel=c * 1, num_cls=num_cls)

    def fuse_levels(self, x1, x2, x3, x4):
        """the fusion blocks of levels 4..1 (reference mmformer.py:159-176: RFM4..RFM1 on the four modality features): they depend
        on the encoders only, so Model.forward may run them on a side stream while the transformers occupy the main one"""
        xs = (x1, x2, x3, x4)
        return {lvl: getattr(self, f"RFM{lvl}")(xs[lvl - 1]) for lvl in (4, 3, 2, 1)}

    def forward(self, x1, x2, x3, x4, x5, mask=None, fused=None, wait=None, on_pred=None):
        """x1..x4: lists of the four modality Feats (already carrying their masks); x5: fused level-5 Feat.
        fused: fuse_levels(x1..x4) computed beforehand (on another stream: `wait()` is called before their first use)"""
        xs = (x1, x2, x3, x4)
        feats, preds = {}, {}
        feats[5] = self.RFM5([x5], want_stats=False)        # feeds a logit head and an upsampling only
        preds[4] = run_conv(self.seg_d4, feats[5], norm=False, out_dtype=torch.float32).t
        if on_pred is not None:                       # (Model.forward: the level's deep-supervision loss starts now, on its side stream)
            on_pred(4, preds[4])
        h = self.d4_c1(_up2(feats[5]))
        if wait is not None:
            wait()
        for lvl in (4, 3, 2, 1):
            r = fused[lvl] if fused is not None else getattr(self, f"RFM{lvl}")(xs[lvl - 1])
            f = getattr(self, f"d{lvl}_out")(getattr(self, f"d{lvl}_c2")([r, h]), want_stats=False)
            feats[lvl] = f
            if lvl > 1:
                preds[lvl - 1] = run_conv(getattr(self, f"seg_d{lvl - 1}"), f, norm=False, out_dtype=torch.float32).t
                if on_pred is not None:
                    on_pred(lvl - 1, preds[lvl - 1])
                h = getattr(self, f"d{lvl - 1}_c1")(_up2(f))
        logits = run_conv(self.seg_layer, feats[1], norm=False, out_dtype=torch.float32).t
        return logits, (preds[1], preds[2], preds[3], preds[4]), tuple(feats[i].t for i in (1, 2, 3, 4, 5))


class Linear(nn.Linear):
    """nn.Linear of the token GEMMs: bf16 operand images kept across calls (ops.bf16_param) under bf16 autocast, and the bias
    gradient from the library's fixed-order column sum (ops.LinearFn) in either precision"""

    def forward(self, x):
        if not self.weight.is_cuda:
            return F.linear(x, self.weight, self.bias)
        bf16 = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if bf16 and x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        return ops.linear(x, self.weight, self.bias, bf16)


class SelfAttention(nn.Module):
    """reference mmformer.py:192-227; the softmax(QK^T)V core is the HIP attention op"""

    def __init__(self, dim, heads=8, qkv_bias=False, qk_scale=None, dropout_rate=0.0):
        super().__init__()
        self.num_heads = heads
        head_dim = dim // heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(dropout_rate)
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(dropout_rate)

    def forward(self, x, drop=True):
        qkv = self.qkv(x)
        p = self.attn_drop.p if self.training else 0.0
        x = ops.attention(qkv, self.num_heads, self.scale, p)
        x = self.proj(x)
        return self.proj_drop(x) if drop else x          # drop=False: the caller applies proj_drop together with what follows (Residual)


def _autocast_bf16():
    """bf16 when the token GEMMs run under bf16 autocast (their Linear casts its input then: the LayerNorm in front writes it directly)"""
    return torch.bfloat16 if (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16) else None


FUSE_TOKEN_GLUE = _os.environ.get("PASSION_FUSE_TOKEN_GLUE", "1") != "0"   # dropout / GELU / residual adds of the transformer blocks in one launch each (ops.token_drop)


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        f = self.fn
        if FUSE_TOKEN_GLUE and self.training and x.is_cuda and x.dtype == torch.float32:
            # training on the GPU: the block's dropouts and its residual add in one pass (the reference's op sequence, mmformer.py:247,256,226 /
            # :247,262,271-275, with the masks of its nn.Dropout modules drawn by the kernel)
            if isinstance(f, PreNormDrop) and isinstance(f.fn, SelfAttention):
                y = f.fn(_ln(f.norm, x, out_dtype=_autocast_bf16()), drop=False)
                if f.fn.proj_drop.p == f.dropout.p and ops.token_drop_ok(y, x):
                    return ops.token_drop(y, x, f.dropout.p, 2, 0)
                return f.dropout(f.fn.proj_drop(y)) + x
            if isinstance(f, PreNorm) and isinstance(f.fn, FeedForward):
                return f.fn(_ln(f.norm, x, out_dtype=_autocast_bf16()), residual=x)
        return self.fn(x) + x


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x):
        return self.fn(_ln(self.norm, x, out_dtype=_autocast_bf16()))


class PreNormDrop(nn.Module):
    def __init__(self, dim, dropout_rate, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.fn = fn

    def forward(self, x):
        return self.dropout(self.fn(_ln(self.norm, x, out_dtype=_autocast_bf16())))


class GELU(nn.Module):
    def forward(self, x):
        return F.gelu(x)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout_rate):
        super().__init__()
        self.net = nn.Sequential(Linear(dim, hidden_dim), GELU(), nn.Dropout(p=dropout_rate),
                                 Linear(hidden_dim, dim), nn.Dropout(p=dropout_rate))

    def forward(self, x, residual=None):
        """residual given (training on the GPU, Residual.forward): -> residual + net(x), GELU + dropout and dropout + add as one launch each"""
        if residual is None:
            return self.net(x)
        l1, _, d1, l2, d2 = self.net
        h = l1(x)
        h = ops.token_drop(h, None, d1.p, 1, 1) if ops.token_drop_ok(h) else d1(F.gelu(h))
        h = l2(h)
        return ops.token_drop(h, residual, d2.p, 1, 0) if ops.token_drop_ok(h, residual) else d2(h) + residual


class Transformer(nn.Module):
    """reference mmformer.py:282-313 (token GEMMs / LayerNorm / GELU stay on torch = hipBLASLt)"""

    def __init__(self, embedding_dim, depth, heads, mlp_dim, dropout_rate=0.1, n_levels=1, n_points=4):
        super().__init__()
        att, ffn = [], []
        self.depth = depth
        for _ in range(depth):
            att.append(Residual(PreNormDrop(embedding_dim, dropout_rate,
                                            SelfAttention(embedding_dim, heads=heads, dropout_rate=dropout_rate))))
            ffn.append(Residual(PreNorm(embedding_dim, FeedForward(embedding_dim, mlp_dim, dropout_rate))))
        self.cross_attention_list = nn.ModuleList(att)
        self.cross_ffn_list = nn.ModuleList(ffn)

    def forward(self, x, pos):
        for j in range(self.depth):
            x = x + pos
            x = self.cross_attention_list[j](x)
            x = self.cross_ffn_list[j](x)
        return x


class MaskModal(nn.Module):
    """kept for tree/API fidelity (reference mmformer.py:316-325); the product forward never copies:
    masks travel with the Feats into the conv loaders"""

    def forward(self, x, mask):
        B, K, C, Hh, Ww, Zz = x.size()
        return (x * mask.view(B, K, 1, 1, 1, 1).to(x.dtype)).view(B, -1, Hh, Ww, Zz)


class Model(nn.Module):
    def __init__(self, num_cls=4):
        super().__init__()
        self.flair_encoder = Encoder()
        self.t1ce_encoder = Encoder()
        self.t1_encoder = Encoder()
        self.t2_encoder = Encoder()

        td = transformer_basic_dims
        for m in MODALS:
            setattr(self, f"{m}_encode_conv", nn.Conv3d(basic_dims * 16, td, kernel_size=1, stride=1, padding=0))
        for m in MODALS:
            setattr(self, f"{m}_pos", nn.Parameter(torch.zeros(1, n_tokens(patch_size), td)))
        for m in MODALS:
            setattr(self, f"{m}_transformer", Transformer(embedding_dim=td, depth=depth, heads=num_heads, mlp_dim=mlp_dim))
        self.multimodal_transformer = Transformer(embedding_dim=td, depth=depth, heads=num_heads, mlp_dim=mlp_dim,
                                                  n_levels=num_modals)
        self.multimodal_decode_conv = nn.Conv3d(td * num_modals, basic_dims * 16 * num_modals, kernel_size=1, padding=0)

        eye = np.eye(4, dtype=bool)
        self.masks_mod0 = torch.from_numpy(eye[0:1].copy())
        self.masks_mod1 = torch.from_numpy(eye[1:2].copy())
        self.masks_mod2 = torch.from_numpy(eye[2:3].copy())
        self.masks_mod3 = torch.from_numpy(eye[3:4].copy())

        self.decoder_fuse = Decoder_fuse(num_cls=num_cls)
        self.decoder_sep = Decoder_sep(num_cls=num_cls)

        self.up2 = nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
        self.up4 = nn.Upsample(scale_factor=4, mode="trilinear", align_corners=True)
        self.up8 = nn.Upsample(scale_factor=8, mode="trilinear", align_corners=True)
        self.up16 = nn.Upsample(scale_factor=16, mode="trilinear", align_corners=True)
        self.up_ops = nn.ModuleList([self.up2, self.up4, self.up8, self.up16])

        self.masker = MaskModal()
        self.mask_type = "idt"
        self.use_passion = False
        self.is_training = False
        self.compute_dtype = torch.float32   # torch.bfloat16 = the bench's storage dtype (f32 accumulate)
        self.num_cls = num_cls
        self.token_dim = td

        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                torch.nn.init.kaiming_normal_(m.weight)

    # ------------------------------------------------------------------------------------------
    def _transformer(self, tr, tok, pos):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.compute_dtype == torch.bfloat16):
            return tr(tok, pos)

    def _inter(self, intra, tok_mask, pos_cat, B, p, P=1):
        """concat the (masked) intra tokens modality-major -> inter-modal transformer -> reinterpret
        (B,4N,512) as a (B,p,p,p,2048) channels-last volume [the reference's .view, mmformer.py:444,
        is this very reinterpretation] -> 1x1x1 decode conv.  With P passes (teacher + students) the token
        sets of all passes go through the transformer as one batch of P*B: tok_mask is (P*B, 4)."""
        if tok_mask is None:
            toks = intra
        else:
            toks = [(t.unsqueeze(0) * tok_mask[:, i].view(P, B, 1, 1).to(t.dtype)).view(P * B, t.shape[1], t.shape[2])
                    for i, t in enumerate(intra)]
        tok = self._transformer(self.multimodal_transformer, torch.cat(toks, 1), pos_cat)
        vol = tok.to(self.compute_dtype).contiguous().view(P * B, *p, self.token_dim * num_modals)
        return run_pointwise(self.multimodal_decode_conv, Feat(vol))

    def _enc_streams(self, dev, vox=0, passion=False):
        """the four side streams of this model on `dev` (decoder_sep, the fusion blocks, the loss branches, the early optimizer pass).
        They are TWO HIP streams for volumes of 128^3 and more, four below (PASSION_SIDE_STREAMS=n: n): with the main stream that is
        three, resp. five, streams on the runtime's four hardware queues -- at 128^3 every kernel of the two finest levels fills the chip
        and a third or fourth side stream only adds queue switches (plain step +0.7 %, the N > 1 code path +3.4 %), at 80^3 the
        kernels are small enough for the concurrency to pay (-1.8 % with two).  Decided by the first forward on the device."""
        pools = self.__dict__.setdefault("_stream_pool", {})
        if dev not in pools:
            env = _os.environ.get("PASSION_SIDE_STREAMS")
            n = max(1, min(4, int(env))) if env else (2 if vox >= 3 << 19 else 4)
            pools[dev] = [torch.cuda.Stream(device=dev) for _ in range(n)]
        pool = pools[dev]
        n = len(pool)
        smap = _os.environ.get("PASSION_SIDE_MAP")                 # A/B: which of the n streams each of the four roles takes, e.g. 0110
        # Two streams, plain step (round 6): the deep-supervision losses share the FUSION blocks' stream, not decoder_sep's -- their backward
        # is what the fusion decoder's chain on the main stream waits for, level by level, and on decoder_sep's stream it sat behind that
        # branch's chip-filling 4 B-sample launches (with decoder_sep's loss created first, below): 9.92 -> 9.69 ms per replayed step, 5 of 5
        # runs (the old map under the new order: bimodal, 9.67 / 10.2).  The PASSION step keeps the old map (measured: 16.87 against 17.11)
        default = [0, 1, 1, 0] if (n == 2 and not passion) else [0, 1, 2, 3]
        roles = [pool[int(c) % n] for c in smap] if smap and len(smap) == 4 else [pool[i % n] for i in default]
        self.__dict__.setdefault("_streams_dev", {})[dev] = roles      # (train._early_step_callback reads the roles of the last forward)
        return roles

    def _deep_weights_on(self, dev):
        cache = self.__dict__.setdefault("_deep_w_dev", {})
        if dev not in cache:
            cache[dev] = torch.tensor([0.5, 0.25, 0.125, 0.0625], device=dev).view(4, 1, 1)
        return cache[dev]

    def _onehots_on(self, dev):
        """masks_mod0..3 (plain CPU attributes, as in the reference) copied to the device once, not per forward"""
        cache = self.__dict__.setdefault("_onehots_dev", {})
        if dev not in cache:
            cache[dev] = [getattr(self, f"masks_mod{i}").to(dev) for i in range(4)]
        return cache[dev]

    @property
    def module(self):
        """the reference reaches the model through DataParallel's `.module` (code/train.py:91-92,212: model.module.mask_type = ...);
        with one process per GPU there is no wrapper, the model answers for it"""
        return self

    def forward(self, x, mask, target=None, temp=1.0):
        L.require_gpu(x)
        ops.set_pack_f32(self.compute_dtype != torch.bfloat16)      # a bf16 model never reads the f32 operand images of its weight packs
        ops.repack_stale()       # all conv weights the optimizer touched, one launch
        ops.recast_stale()       # ... and the bf16 images of the token-GEMM parameters, one more
        B, S = x.size(0), x.size(2)
        dev = x.device
        dt = self.compute_dtype
        idt = self.mask_type != "pdt"
        training = self.is_training
        passion = training and self.use_passion
        mask = mask.to(device=dev, dtype=torch.bool).contiguous()
        grouped = GROUPED_ENC and x.is_cuda and L.host() is not None
        streams = self._enc_streams(dev, x.shape[2] * x.shape[3] * x.shape[4], passion) if ENC_STREAMS and x.is_cuda else None
        main = torch.cuda.current_stream(dev) if streams else None
        if streams:
            for s_ in streams:                           # (possibly a fresh copy made on the main stream) read on every side stream
                mask.record_stream(s_)
        encs = [None] * 4
        intra = [None] * 4
        pos = [getattr(self, f"{m}_pos") for m in MODALS]
        stacked_feats = None
        # the label volume in the losses' layout: it depends on the batch alone, so its conversion (a 33 MB strided copy that sat between
        # the end of the forward and the first loss kernel: 37 us with nothing beside it) goes to a side stream at the very top
        tcl, tcl_ev, gate_early = None, None, None
        if training and target is not None and streams and LOSS_STREAMS and TCL_EARLY:
            s2 = streams[2]
            s2.wait_stream(main)
            with torch.cuda.stream(s2):
                tcl = ops.target_to_cl(target)
                gate_early = mask.float() if idt else torch.ones(B, 4, device=dev)
                tcl_ev = s2.record_event()
            tcl.record_stream(main)
            gate_early.record_stream(main)

        def intra_tokens(i):
            """token conv + intra-modal transformer of modality i (mmformer.py:417-433): a chain of ~30 tiny launches that
            depends on encoder i only, so it stays on that encoder's stream"""
            m = MODALS[i]
            own5 = encs[i][4].with_mask(mask, i) if idt else encs[i][4]
            tok = run_pointwise(getattr(self, f"{m}_encode_conv"), own5).t.view(B, -1, self.token_dim)
            tok = self._transformer(getattr(self, f"{m}_transformer"), tok, pos[i])
            return tok * mask[:, i].view(B, 1, 1).to(tok.dtype)

        if grouped:
            # ---- the four encoders as grouped launches over the stacked batch (encoders_grouped); the per-modality tensors the
            # fusion path reads are the four batch slices of every level (ops.unstack: shared storage, no copy either way)
            stacked_feats = encoders_grouped([getattr(self, f"{m}_encoder") for m in MODALS], x, mask if idt else None, dt)
            if training:                               # tells train_step when only the encoders' backward is left (ops.BackwardMarkerFn)
                f5 = stacked_feats[4]
                stacked_feats[4] = Feat(ops.backward_marker(f5.t), f5.mask, f5.mask_col, _stats=f5._stats)
            per_level = []
            for l, f in enumerate(stacked_feats):
                ts = ops.unstack(f.alt if f.alt is not None else f.t, 4)      # (the fusion path's own handed-on copy of the level: encoders_grouped)
                st = f._stats[0].view(4, B, -1, 2) if f._stats[0] is not None else None
                per_level.append([Feat(ts[i], _stats=[st[i]] if st is not None else None) for i in range(4)])
            encs = [[per_level[l][i] for l in range(5)] for i in range(4)]
            if GROUPED_INTRA:                            # the four intra-modal transformers as batched GEMMs over the stack
                toks = intra_grouped(self, stacked_feats[4].t, mask, B, dt == torch.bfloat16)
                intra = list(ops.unstack(toks.view(4 * B, toks.shape[2], toks.shape[3]), 4))
            elif streams and INTRA_STREAMS:              # ... or one by one on the side streams
                for i in range(4):
                    streams[i].wait_stream(main)
                    encs[i][4].t.record_stream(streams[i])
                    with torch.cuda.stream(streams[i]):
                        intra[i] = intra_tokens(i)
                        intra[i].record_stream(main)
        else:
            xin = x.to(dt).contiguous()
            # ---- encoders (idt: input and features of absent modalities are zero, mmformer.py:397-415).
            # When training, the level outputs of the four modalities land back to back in one buffer per level:
            # decoder_sep (shared weights) then runs ONCE over the 4B stack instead of four times.
            bufs = sbufs = None
            if training:
                c = self.flair_encoder.e1_c1.out_channels
                Dv, Hv, Wv = x.shape[2:]              # any (16 pD, 16 pH, 16 pW) volume, not only cubes
                bufs = [torch.empty(4, B, Dv >> l, Hv >> l, Wv >> l, c << l, dtype=dt, device=dev) for l in range(5)]
                sbufs = [torch.empty(4, B, c << l, 2, dtype=torch.float32, device=dev) for l in range(5)]
            # The four encoders are independent until the fusion: each runs on its own HIP stream (forward, and -- autograd
            # replays every node on the stream of its forward -- backward too), so the small deep-level launches of one
            # modality fill the CUs the others leave idle.  PASSION_ENC_STREAMS=0 keeps everything on one stream.

            for i, m in enumerate(MODALS):
                outs = [ops.alias_slice(b_, i) for b_ in bufs] if training else None
                sts = [s_[i] for s_ in sbufs] if training else None
                if streams:
                    streams[i].wait_stream(main)
                    if training:
                        for b_ in bufs + sbufs:              # allocated on the main stream, written and (backward) read on this one
                            b_.record_stream(streams[i])
                    with torch.cuda.stream(streams[i]):
                        encs[i] = getattr(self, f"{m}_encoder")(xin, i, mask if idt else None, dt, outs, sts)
                        for f in encs[i]:                    # consumed by work on the main stream: tell the allocator, or the
                            f.t.record_stream(main)          # block could be handed out again on this side stream too early
                            if f._stats[0] is not None:
                                f._stats[0].record_stream(main)
                        if INTRA_STREAMS:
                            intra[i] = intra_tokens(i)
                            intra[i].record_stream(main)
                else:
                    encs[i] = getattr(self, f"{m}_encoder")(xin, i, mask if idt else None, dt, outs, sts)
        if streams:
            for s_ in streams:
                main.wait_stream(s_)
        p = tuple(encs[0][4].t.shape[1:4])       # token grid (pD, pH, pW); the reference's is the cube patch_size^3

        # ---- the masks of every pass through the fusion path: teacher (+ the four uni-modal students,
        # mmformer.py:468-531).  All P passes share the encoder features, so they run as ONE virtual batch of
        # P*B: each conv that reads encoder features re-reads sample b % B under mask row b.
        P = 1
        masks_all, tok_mask = mask, None
        if passion:
            P = 5
            onehots = [t.expand(B, 4) for t in self._onehots_on(dev)]
            # idt: absent-by-data modalities stay zero as well -- the student sees mask AND one-hot
            studs = [(mask & onehots[i]) if idt else onehots[i] for i in range(4)]
            masks_all = torch.cat([mask] + studs, 0).contiguous()
            tok_mask = torch.cat([torch.ones_like(mask)] + [onehots[2 if i == 3 else i] for i in range(4)], 0)  # quirk :522
        xs = [[encs[i][l].with_mask(masks_all, i, P) for i in range(4)] for l in range(4)]

        # ---- intra-modal transformers (mmformer.py:417-433), unless they already ran on the encoder streams
        for i in range(4):
            if intra[i] is None:
                intra[i] = intra_tokens(i)
        pos_cat = torch.cat(pos, 1)

        # decoder_sep (per-modality decoders, shared weights: ONE pass over the stacked modalities) depends on the encoders
        # only, like the fusion path: it goes to a side stream and overlaps with the transformers + decoder_fuse below
        sep_all = None
        if training:
            sep_mask = mask.t().reshape(4 * B, 1).contiguous() if idt else None      # row i*B+b = mask[b, i]
            stacked = []
            for l in range(5):
                if grouped:                              # the grouped encoders' outputs ARE the stack
                    f = stacked_feats[l]
                    stacked.append(Feat(f.t, sep_mask, 0, _stats=f._stats))
                    continue
                t = ops.stack_slots(bufs[l], [encs[i][l].t for i in range(4)])
                st = [sbufs[l].view(4 * B, -1, 2)] if l < 4 else None               # filled by the encoders' next-level convs
                stacked.append(Feat(t, sep_mask, 0, _stats=st))
            if streams:
                streams[0].wait_stream(main)
                if sep_mask is not None:
                    sep_mask.record_stream(streams[0])  # made on the main stream a moment ago, read by every conv of decoder_sep
                if grouped:                              # allocated on the main stream, read (forward and backward) on this one
                    for f in stacked:
                        f.t.record_stream(streams[0])
                        if f._stats[0] is not None:
                            f._stats[0].record_stream(streams[0])
                with torch.cuda.stream(streams[0]):
                    sep_all = self.decoder_sep(*stacked, probs=not ops.SEP_LOSS_FUSED)
                    sep_all.record_stream(main)
            else:
                sep_all = self.decoder_sep(*stacked, probs=not ops.SEP_LOSS_FUSED)

        # The fusion blocks of levels 4..1 read encoder features only: on a side stream they run under the transformers' chain of
        # tiny launches (forward) and under the decoder's own backward chain (autograd replays them on that stream)
        fused = wait = None
        if streams and FUSE_STREAM:
            s1 = streams[1]
            s1.wait_stream(main)
            masks_all.record_stream(s1)
            for lv in xs:
                for f in lv:                              # made on the main stream, read (forward and backward) on this one
                    f.t.record_stream(s1)
                    if f._stats[0] is not None:
                        f._stats[0].record_stream(s1)
            with torch.cuda.stream(s1):
                fused = self.decoder_fuse.fuse_levels(*xs)
                for f in fused.values():
                    f.t.record_stream(main)
                    if f._stats[0] is not None:
                        f._stats[0].record_stream(main)
            wait = lambda: main.wait_stream(s1)           # noqa: E731
        # deep supervision EARLY: a level's logits exist long before the forward ends; softmax + loss sums of that level run on the loss
        # stream as soon as they do (on a detached copy: ops.DeferredFn attaches the little graph later, where the node used to be
        # created, so the backward's order does not change) instead of as a chain of four after the last conv of the forward
        early_ds = {}
        on_pred = None
        if tcl is not None and DS_EARLY:
            def on_pred(l, pp):
                s2 = streams[2]
                s2.wait_stream(main)
                pp.record_stream(s2)
                with torch.cuda.stream(s2):
                    xd = pp[:B].detach().requires_grad_()
                    early_ds[l] = [crit.seg_losses_cl(ops.softmax_classes(xd, 1.0, "cl"), tcl, 2 ** l), xd]
        fuse_all, preds_all, feats_all = self.decoder_fuse(*xs, self._inter(intra, tok_mask, pos_cat, B, p, P), fused=fused, wait=wait,
                                                           on_pred=on_pred)
        if P == 1:
            fuse_pred, preds = fuse_all, preds_all
        else:
            # rows [0, B): the teacher; rows [B, 5B): the four uni-modal students, student-major
            fuse_pred, fuse_stud = fuse_all.split([B, 4 * B], 0)
            preds_ts = [pp.split([B, 4 * B], 0) for pp in preds_all]
            preds, preds_stud = [t[0] for t in preds_ts], [t[1] for t in preds_ts]
            f1_teacher, f1_stud = feats_all[0].split([B, 4 * B], 0)
        pred = ops.softmax_classes(fuse_pred, 1.0, "ncdhw")
        if not training:
            return pred

        if tcl is None:
            tcl = ops.target_to_cl(target)
        else:
            main.wait_event(tcl_ev)
            if DS_EARLY:
                # the fusion prediction's own loss (train.assemble_loss forms it: reference train.py:228) on a detached copy NOW, on the main
                # stream, while the side streams finish theirs -- it used to follow decoder_sep's loss and the joins; the node is attached
                # where assemble_loss asks for the loss (utils/criterions._seg_pair), so the backward's order stays
                xd = pred.detach().requires_grad_()
                pred._passion_pair_box = (tcl, [ops.wce_dice(xd, tcl, "ncdhw"), xd])
        # Loss glue with as few tiny launches as possible (each is ~4 us of dependent latency on the main stream between forward and
        # backward, and twice that in host work): (B, 4) matrices instead of per-modality loops.  gate[b, i] = 1 if modality i of
        # sample b takes part (idt: present in the data; pdt: always) -- the reference's `* mask` / one-hot selection.
        gate = gate_early if gate_early is not None else (mask.float() if idt else torch.ones(B, 4, device=dev))

        def deep_supervision():
            # softmax -> upsample -> WCE + Dice (mmformer.py:540-544), weights 1/2, 1/4, 1/8, 1/16
            pairs = [ops.DeferredFn.apply(pp, early_ds[j + 1]) if (j + 1) in early_ds
                     else crit.seg_losses_cl(ops.softmax_classes(pp, 1.0, "cl"), tcl, 2 ** (j + 1)) for j, pp in enumerate(preds)]
            return (torch.stack(pairs, 0) * self._deep_weights_on(dev)).sum((0, 2)).view(B, 1)

        def sep_losses():
            # the four per-modality predictions (4B samples, modality-major) against the B label volumes in one launch
            return gate * ops.sep_wce_dice(sep_all, tcl).sum(1).view(4, B).t()

        if streams and LOSS_STREAMS:
            # Both are side branches of the graph: on side streams their backward (softmax / upsample / loss gradients) runs under
            # the start of the fusion decoder's backward chain instead of in front of it.  decoder_sep's loss stays on decoder_sep's
            # stream, the deep-supervision branch gets its own.
            s0, s2 = streams[0], streams[2]
            for s_ in (s0, s2):
                # decoder_sep's loss needs decoder_sep's logits (its own stream), the labels and the gate (made at the top of the forward
                # on the loss stream: an event) -- not the end of the fusion decoder's forward, which it used to wait for here
                if s_ is s0 and s0 is not s2 and tcl_ev is not None and DS_EARLY:      # (PASSION's stream map gives both roles one stream: wait for main)
                    s_.wait_event(tcl_ev)
                else:
                    s_.wait_stream(main)
                tcl.record_stream(s_)
            gate.record_stream(s0)
            for pp in preds:
                pp.record_stream(s2)
            # decoder_sep's loss FIRST (round 6): autograd runs ready nodes in reverse creation order, so the deep-supervision backward --
            # which the fusion decoder's chain on the main stream waits for, level by level -- is then enqueued AHEAD of decoder_sep's
            # loss gradient (a chip-filling 4 B-sample launch) instead of behind it on the stream the two share: 9.94 -> 9.72 ms per plain
            # step (3 x 100 replayed steps each; PASSION +-0).  PASSION_SEP_LOSS_FIRST=0: the old order
            if SEP_LOSS_FIRST:
                with torch.cuda.stream(s0):
                    sep_loss = sep_losses()
                    sep_loss.record_stream(main)
            with torch.cuda.stream(s2):
                prm_loss = deep_supervision()
                prm_loss.record_stream(main)
            if not SEP_LOSS_FIRST:
                with torch.cuda.stream(s0):
                    sep_loss = sep_losses()
                    sep_loss.record_stream(main)
            main.wait_stream(s0)
            main.wait_stream(s2)
        else:
            if streams:
                main.wait_stream(streams[0])
            prm_loss, sep_loss = deep_supervision(), sep_losses()
        if not passion:
            return pred, prm_loss, sep_loss

        # ---- PASSION losses of the four students against the (detached) teacher
        inv_t = 1.0 / temp
        pt_full = ops.softmax_classes(fuse_pred.detach(), inv_t, "cl")
        pt_deep = [ops.softmax_classes(pp.detach(), inv_t, "cl") for pp in preds]        # coarse: crit.kl_cl upsamples inside the kernel
        f_teacher = f1_teacher.detach()
        # The four students are rows of ONE batch: every loss runs once over 4B samples against the B teacher samples (the kernels
        # index teacher / target with b % B) instead of four times -- a quarter of the ~190 loss-chain launches of a PASSION step
        pl_all, d_all = ops.proto_loss(f1_stud, f_teacher, tcl)                       # (4B, 1) each, student-major
        kl_all = crit.kl_cl(fuse_stud, pt_full, temp)
        for j in range(4):
            kl_all = kl_all + 0.5 ** (j + 1) * crit.kl_cl(preds_stud[j], pt_deep[j], temp, 2 ** (j + 1))
        # A sample whose only present modality is i feeds student i exactly what the teacher saw, so the reference (deterministic
        # kernels) gets kl = proto = dist = 0 *exactly* -- and train.py:266-267 then turns that 0/0 into an all-false rp_mask.
        # Teacher and student are different rows of one batched launch here (GEMM rounding may differ per row), so the identity
        # is imposed instead of hoped for.  (Not for T2: its tokens are T1's, mmformer.py:522.)
        # With dropout LIVE (model.train(): the reference's training loop) every pass of the reference draws its own dropout masks in
        # the inter-modal transformer (mmformer.py:468-531), so such a student differs from the teacher and its kl / proto / dist are
        # small positive numbers there, as they are here (teacher and students are different rows of the batched dropout): nothing
        # is imposed then.
        only = mask & (mask.sum(1, keepdim=True) == 1)                                # [b, i]: i is sample b's only modality
        differs = (~only).float()
        differs[:, 3] = 1.0
        if self.training:
            differs = torch.ones_like(differs)
        gd = gate * differs
        kl_loss = gd * kl_all.view(4, B).t()
        proto_loss = gd * pl_all.view(4, B).t()
        dist = gd * d_all.view(4, B).t()
        return pred, prm_loss, sep_loss, kl_loss, proto_loss, dist
