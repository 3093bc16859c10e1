"""Building blocks of the MI355X-native PASSION models.

Mirrors the reference's code/models/blocks.py *interface* for the classes on the hot path
(general_conv3d_prenorm :300-316, fusion_prenorm :533-542): same class names, constructor
signatures and parameter names/shapes, so a reference state_dict loads with strict=True.  The
parameters live in ordinary nn.Conv3d holders (identical default initialisation and RNG order);
the forward never calls ATen convolution -- it launches the fused HIP conv kernel through
passion_amd.ops on channels-last tensors.
"""
import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops

PAD = {"zeros": L.PAD_ZEROS, "reflect": L.PAD_REFLECT}
import os as _os
MATERIALIZE = _os.environ.get("PASSION_MATERIALIZE", "0") == "1"   # experiment: activations normalised once per tensor


class Feat:
    """A channels-last activation (B,D,H,W,C) plus its lazily computed InstanceNorm statistics and
    an optional modality mask (mask tensor, column): consumers treat absent modalities as zeros.
    With `passes` > 1 the mask has passes*B rows and a consuming conv runs all the masked passes
    (teacher + uni-modal students) over the same B samples in one launch (ops.SegIn.bmod).
    `stats_out`: optional preallocated (B,C,2) slot the statistics are written into (one slice of the
    statistics of a stacked buffer)."""
    __slots__ = ("t", "_stats", "mask", "mask_col", "passes", "stats_out", "_act", "c_real", "alt")

    def __init__(self, t, mask=None, mask_col=0, _stats=None, passes=1, stats_out=None, _act=None, c_real=None):
        self.t, self.mask, self.mask_col, self.passes, self.stats_out = t, mask, mask_col, passes, stats_out
        self.c_real = c_real     # channels that carry data when the tensor was zero-padded to a multiple of 8 (RFNet)
        self.alt = None          # a second alias of `t` handed on by the conv that reads it (mmformer.encoders_grouped): for another reader
        self._stats = _stats if _stats is not None else [None]   # shared between masked views of one tensor
        self._act = _act if _act is not None else [None]         # LeakyReLU(InstanceNorm(t)) when materialised

    def stats(self):
        if self._stats[0] is None:
            self._stats[0] = ops.instnorm_stats(self.t.detach(), out=self.stats_out)
        return self._stats[0]

    def with_mask(self, mask, col, passes=1):
        return Feat(self.t, mask, col, self._stats, passes, self.stats_out, self._act, self.c_real)

    def act(self):
        if self._act[0] is None:
            self._act[0] = ops.norm_act(self.t, self.stats())
        return self._act[0]

    def seg(self, norm=True):
        bmod = self.t.shape[0] if self.passes > 1 else 0
        if norm and MATERIALIZE:      # the conv loaders then copy instead of normalising every halo voxel again
            return ops.SegIn(self.act(), stats=None, mask=self.mask, mask_col=self.mask_col, bmod=bmod)
        return ops.SegIn(self.t, stats=self.stats() if norm else None, mask=self.mask, mask_col=self.mask_col, bmod=bmod)

    @property
    def dims(self):
        return self.t.shape[1], self.t.shape[2], self.t.shape[3]


def normalization(planes, norm="in"):
    """only the 'in' variant is ever requested on the hot path (reference blocks.py:12-23)"""
    if norm == "in":
        return nn.InstanceNorm3d(planes)
    if norm == "bn":
        return nn.BatchNorm3d(planes)
    if norm == "gn":
        return nn.GroupNorm(4, planes)
    raise ValueError(f"normalization type {norm} is not supported")


FUSE_STATS = _os.environ.get("PASSION_FUSE_STATS", "1") != "0"   # A/B: 0 = statistics by their own launches, on first use


def run_conv(conv, feats, norm, residual=None, out_dtype=None, out=None, stats_out=None, want_stats=False):
    """Launch the fused conv for an nn.Conv3d parameter holder over the virtual concat of `feats`.
    want_stats: the output goes on into a pre-norm conv -- have this conv leave the InstanceNorm statistics of its output too
    (from its own epilogue where the kernel can, ops.conv_block) instead of a separate pass on first use."""
    if not isinstance(feats, (list, tuple)):
        feats = [feats]
    D, H, W = feats[0].dims
    cache = conv.__dict__.setdefault("_passion_pack", {})
    want_stats = want_stats and FUSE_STATS and not MATERIALIZE
    y = ops.conv_block([f.seg(norm) for f in feats], conv.weight, conv.bias, D, H, W,
                       ksize=conv.kernel_size[0], stride=conv.stride[0], pad_mode=PAD[conv.padding_mode],
                       residual=residual, out_dtype=out_dtype, pack_cache=cache, out=out, want_stats=want_stats,
                       stats_out=stats_out)
    if want_stats:
        return Feat(y[0], _stats=[y[1]], stats_out=stats_out)
    return Feat(y, stats_out=stats_out)


GEMM_1X1 = _os.environ.get("PASSION_GEMM_1X1", "1") != "0"


def run_pointwise(conv, feat):
    """A 1x1x1 conv WITHOUT a normalisation in front (the token convs *_encode_conv and multimodal_decode_conv, reference
    mmformer.py:213-214, 243) is a plain GEMM over the voxels of a channels-last tensor: it goes to hipBLASLt like the other
    token GEMMs (ops.linear; a 2048 -> 512 conv over 512 voxels took 180 us per direction in the general conv kernel).  The
    modality mask of `feat` is not applied: both call sites multiply the result's consumers by that mask anyway."""
    t = feat.t
    if not (GEMM_1X1 and t.is_cuda and conv.kernel_size[0] == 1 and conv.stride[0] == 1):
        return run_conv(conv, feat, norm=False)
    B, D, H, W, C = t.shape
    y = ops.linear(t.view(B, D * H * W, C), conv.weight, conv.bias, t.dtype == torch.bfloat16)
    return Feat(y.view(B, D, H, W, conv.out_channels))


class general_conv3d_prenorm(nn.Module):
    """InstanceNorm -> LeakyReLU(0.2) -> Conv3d, one fused kernel (reference blocks.py:300-316).
    Note the reference builds InstanceNorm3d(out_ch) but applies it to the in_ch input; with
    affine=False that is harmless and reproduced here only as an attribute."""

    def __init__(self, in_ch, out_ch, k_size=3, stride=1, padding=1, pad_type="zeros", norm="in", is_training=True,
                 act_type="lrelu", relufactor=0.2):
        super().__init__()
        self.conv = nn.Conv3d(in_channels=in_ch, out_channels=out_ch, kernel_size=k_size, stride=stride,
                              padding=padding, padding_mode=pad_type, bias=True)
        self.norm = normalization(out_ch, norm=norm)
        self.activation = nn.LeakyReLU(negative_slope=relufactor, inplace=True) if act_type == "lrelu" else nn.ReLU(inplace=True)

    def forward(self, feats, residual=None, out=None, stats_out=None, want_stats=True):
        """want_stats=False: nobody normalises this output (it only feeds an upsampling / a logit head / the token conv)"""
        return run_conv(self.conv, feats, norm=True, residual=residual, out=out, stats_out=stats_out, want_stats=want_stats)


class fusion_prenorm(nn.Module):
    """1x1x1 -> 3x3x3 (zero padded) -> 1x1x1, each pre-normed (reference blocks.py:533-542).
    The first conv reads the four modality feature maps as four masked segments: the reference's
    torch.stack + MaskModal copy (mmformer.py:316-325) never materialises."""

    def __init__(self, in_channel=64, num_cls=4):
        super().__init__()
        self.fusion_layer = nn.Sequential(
            general_conv3d_prenorm(in_channel * num_cls, in_channel, k_size=1, padding=0, stride=1),
            general_conv3d_prenorm(in_channel, in_channel, k_size=3, padding=1, stride=1),
            general_conv3d_prenorm(in_channel, in_channel, k_size=1, padding=0, stride=1))

    def forward(self, feats, want_stats=True):
        h = self.fusion_layer[0](feats)
        h = self.fusion_layer[1](h)
        return self.fusion_layer[2](h, want_stats=want_stats)


# ------------------------------------------------------------------------------------------------
# RFNet (post-norm) blocks
# ------------------------------------------------------------------------------------------------
class PFeat(Feat):
    """raw output of a post-norm general_conv3d: its VALUE is LeakyReLU(InstanceNorm(t)).  A consuming
    conv applies that in its loader (same fused path as the pre-norm blocks); anything else calls
    materialize()."""
    __slots__ = ()


def _seg_auto(f):
    return f.seg(norm=isinstance(f, PFeat))


def materialize(f):
    """Feat whose tensor holds the activated values"""
    if isinstance(f, PFeat):
        return Feat(ops.norm_act(f.t, f.stats()), f.mask, f.mask_col, c_real=f.c_real)
    return f


def _pad_axis(t, dim, mult=4):
    n = t.shape[dim]
    if n % mult == 0:
        return t
    pad = [0, 0] * t.dim()
    pad[2 * (t.dim() - 1 - dim) + 1] = mult - n % mult
    return torch.nn.functional.pad(t, pad)


def run_conv_auto(conv, feats, out_dtype=None, post=False):
    """like run_conv, but every input decides by its own type whether the loader normalises it.  Channel counts of
    bf16 activations that are not multiples of 8 (RFNet's C//4 embeddings and C//2 fusion outputs at the two finest
    levels: 2 / 4 channels) are zero-padded to 8 so that they stay on the matrix-core kernels (the 4-channel versions
    fell back to the f32 general path: 0.9 + 1.6 ms per conv at 128^3): padded output channels are exactly 0 before and
    after InstanceNorm + LeakyReLU and meet zero weight columns downstream; the weight gradient of the padding is
    sliced away by autograd.  A Feat remembers how many of its channels are real (`c_real`)."""
    if not isinstance(feats, (list, tuple)):
        feats = [feats]
    D, H, W = feats[0].dims
    have = [f.t.shape[-1] for f in feats]
    real = [f.c_real or h for f, h in zip(feats, have)]
    w, b = conv.weight, conv.bias
    assert sum(real) == w.shape[1], (real, w.shape)
    if real != have:                    # a producer was padded: pad the matching input-channel ranges with zero columns
        parts, o = [], 0
        for r, h in zip(real, have):
            blk = w[:, o:o + r]
            parts.append(blk if r == h else torch.nn.functional.pad(blk, (0, 0, 0, 0, 0, 0, 0, h - r)))
            o += r
        w = torch.cat(parts, 1) if len(parts) > 1 else parts[0]
    odt = out_dtype if out_dtype is not None else feats[0].t.dtype
    mult = 8 if odt == torch.bfloat16 else 4
    cout = w.shape[0]
    w = _pad_axis(w, 0, mult)
    b = _pad_axis(b, 0, mult) if b is not None else None
    padded = w is not conv.weight
    cache = None if padded else conv.__dict__.setdefault("_passion_pack", {})
    # a post-norm conv's output is always normalised by its consumer: the conv leaves the InstanceNorm statistics of what it
    # stores from its own epilogue where the kernel can (ops.conv_block want_stats) instead of two or three launches on first use
    want = post and FUSE_STATS and feats[0].t.is_cuda
    y = ops.conv_block([_seg_auto(f) for f in feats], w, b, D, H, W, ksize=conv.kernel_size[0], stride=conv.stride[0],
                       pad_mode=PAD[conv.padding_mode], out_dtype=out_dtype, pack_cache=cache, want_stats=want)
    st = None
    if want:
        y, st = y
    c_real = cout if y.shape[-1] != cout else None
    if post:
        return PFeat(y, _stats=[st] if st is not None else None, c_real=c_real)
    return Feat(y, c_real=c_real)


GROUPED_ENC_POST = _os.environ.get("PASSION_GROUPED_ENC", "1") != "0"
RES_FUSE = int(_os.environ.get("PASSION_RES_FUSE", "2"))      # gradient sums inside the conv backward: 0 off; 1 the residual blocks + one handed-on copy of
                                                              # every level's feature; 2: two copies (decoder_sep / fusion path)


def encoders_grouped_post(encs, x, mask, dtype, levels):
    """The four post-norm modality encoders (reference m2ftrans.py:63-104 / rfnet.py:15-48, four Encoder() instances of identical shape) over the stacked
    batch [flair | t1ce | t1 | t2] (4B samples): every conv is ONE grouped launch (ops.conv_block_grouped: sample s uses the
    weights of encoder s // B), the residual x + c3(c2(x)) one norm_act launch -- a quarter of the launches of the per-modality
    loop.  x: (B,4,D,H,W); mask (B,4) bool or None.  -> `levels` Feats of shape (4B, D_l, H_l, W_l, C_l), modality-major."""
    B, K, D, H, W = x.shape
    xs = torch.empty(K, B, D, H, W, 1, dtype=dtype, device=x.device)
    xs.view(K, B, D, H, W).copy_(x.transpose(0, 1))
    xs = xs.view(K * B, D, H, W, 1)
    gmask = mask.t().reshape(K * B, 1).contiguous() if mask is not None else None      # row m * B + b = mask[b, m]

    def gconv(name, f, first=False, pass_x=0):
        convs = [getattr(e, name).conv for e in encs]
        c0 = convs[0]
        d, h, w = f.t.shape[1:4]
        seg = ops.SegIn(f.t, mask=gmask, mask_col=0) if first else f.seg(isinstance(f, PFeat))
        r = ops.conv_block_grouped([seg], [c.weight for c in convs], [c.bias for c in convs], d, h, w, ksize=c0.kernel_size[0],
                                   stride=c0.stride[0], pad_mode=L.PAD_REFLECT, out_dtype=dtype, want_stats=True, pass_x=pass_x)
        if pass_x:
            return (PFeat(r[0], _stats=[r[1]]),) + tuple(r[2:])
        return PFeat(r[0], _stats=[r[1]])

    # RES_FUSE (models/mmformer.py has the pre-norm twin): a tensor with several readers is handed on by the conv that reads it, the others
    # take the handed-on copies, and their gradients join that conv's input gradient inside its backward (passion_seg_t.dx_add / dx_add2)
    # instead of in autograd sums: r1 -> c2 and the residual; a level's feature -> the next level's stride-2 conv, decoder_sep, the fusion path
    r1 = gconv("e1_c1", Feat(xs), first=True)
    feats = []
    for lvl in range(1, levels + 1):
        if lvl > 1 and RES_FUSE >= 2:
            r1, xa, xb = gconv(f"e{lvl}_c1", feats[-1], pass_x=2)
            feats[-1] = Feat(xa)
            feats[-1].alt = xb
        elif lvl > 1:
            r1 = gconv(f"e{lvl}_c1", feats[-1])
        if RES_FUSE:
            y2, r1x = gconv(f"e{lvl}_c2", r1, pass_x=1)
            r3 = gconv(f"e{lvl}_c3", y2)
            feats.append(Feat(ops.norm_act(r1x, r1.stats(), r3.t, r3.stats())))
        else:
            r3 = gconv(f"e{lvl}_c3", gconv(f"e{lvl}_c2", r1))
            feats.append(Feat(ops.norm_act(r1.t, r1.stats(), r3.t, r3.stats())))
    return feats


class general_conv3d(nn.Module):
    """Conv3d -> InstanceNorm -> LeakyReLU(0.2) (reference blocks.py:354-370).  The conv launches here; the
    norm + activation happen in the loader of whichever conv consumes the returned PFeat."""

    def __init__(self, in_ch, out_ch, k_size=3, stride=1, padding=1, pad_type="reflect", norm="in", is_training=True,
                 act_type="lrelu", relufactor=0.2):
        super().__init__()
        self.conv = nn.Conv3d(in_channels=in_ch, out_channels=out_ch, kernel_size=k_size, stride=stride, padding=padding,
                              padding_mode=pad_type, bias=True)
        self.norm = normalization(out_ch, norm=norm)
        self.activation = nn.LeakyReLU(negative_slope=relufactor, inplace=True) if act_type == "lrelu" else nn.ReLU(inplace=True)

    def forward(self, feats):
        return run_conv_auto(self.conv, feats, post=True)


def _chain(seq, feats):
    for layer in seq:
        feats = layer(feats) if isinstance(layer, general_conv3d) else run_conv_auto(layer, feats, out_dtype=torch.float32)
    return feats


class prm_generator_laststage_pk(nn.Module):
    """region probability logits of the deepest level (reference blocks.py:396-416)"""

    def __init__(self, in_channel=64, norm="in", num_cls=4):
        super().__init__()
        q = int(in_channel // 4)
        self.embedding_layer = nn.Sequential(general_conv3d(in_channel * 4, q, k_size=1, padding=0, stride=1),
                                             general_conv3d(q, q, k_size=3, padding=1, stride=1),
                                             general_conv3d(q, in_channel, k_size=1, padding=0, stride=1))
        self.prm_layer = nn.Sequential(general_conv3d(in_channel, 16, k_size=1, stride=1, padding=0),
                                       nn.Conv3d(16, num_cls, kernel_size=1, padding=0, stride=1, bias=True))

    def forward(self, x, mask=None):
        """x: the four (masked) modality Feats"""
        return _chain(self.prm_layer, _chain(self.embedding_layer, x)).t


class prm_generator_pk(nn.Module):
    """reference blocks.py:443-464: prm_layer(cat(x1, embedding_layer(masked x2)))"""

    def __init__(self, in_channel=64, norm="in", num_cls=4):
        super().__init__()
        q = int(in_channel // 4)
        self.embedding_layer = nn.Sequential(general_conv3d(in_channel * 4, q, k_size=1, padding=0, stride=1),
                                             general_conv3d(q, q, k_size=3, padding=1, stride=1),
                                             general_conv3d(q, in_channel, k_size=1, padding=0, stride=1))
        self.prm_layer = nn.Sequential(general_conv3d(in_channel * 2, 16, k_size=1, stride=1, padding=0),
                                       nn.Conv3d(16, num_cls, kernel_size=1, padding=0, stride=1, bias=True))

    def forward(self, x1, x2, mask=None):
        return _chain(self.prm_layer, [x1, _chain(self.embedding_layer, x2)]).t


class modal_fusion(nn.Module):
    """per-region modality weights from global statistics (reference blocks.py:495-517); the tiny
    (B, 4C+1) -> 128 -> 4 MLP stays on torch, the volume work is in region_gap / region_mix"""

    def __init__(self, in_channel=64):
        super().__init__()
        self.weight_layer = nn.Sequential(nn.Conv3d(4 * in_channel + 1, 128, 1, padding=0, bias=True),
                                          nn.LeakyReLU(negative_slope=0.2, inplace=True),
                                          nn.Conv3d(128, 4, 1, padding=0, bias=True))
        self.sigmoid = nn.Sigmoid()

    def weights(self, feat_avg, prm_avg):
        """feat_avg (B, 4*C), prm_avg (B, 1) -> sigmoid weights (B, 4)"""
        w0, w2 = self.weight_layer[0], self.weight_layer[2]
        h = torch.nn.functional.linear(torch.cat((feat_avg, prm_avg), 1), w0.weight.flatten(1), w0.bias)
        h = torch.nn.functional.leaky_relu(h, 0.2)
        return torch.sigmoid(torch.nn.functional.linear(h, w2.weight.flatten(1), w2.bias))


class region_fusion(nn.Module):
    """reference blocks.py:520-531"""

    def __init__(self, in_channel=64, num_cls=4):
        super().__init__()
        self.fusion_layer = nn.Sequential(general_conv3d(in_channel * num_cls, in_channel, k_size=1, padding=0, stride=1),
                                          general_conv3d(in_channel, in_channel, k_size=3, padding=1, stride=1),
                                          general_conv3d(in_channel, in_channel // 2, k_size=1, padding=0, stride=1))

    def forward(self, x):
        return _chain(self.fusion_layer, x)


class region_aware_modal_fusion(nn.Module):
    """reference blocks.py:582-626.  x: the four modality Feats (carrying the mask), prm: (B,D,H,W,4) f32
    region probabilities (already detached).  Returns the two halves of the channel concat
    [region_fusion(...), short_cut(...)] as post-norm Feats."""

    def __init__(self, in_channel=64, norm="in", num_cls=4):
        super().__init__()
        self.num_cls = num_cls
        self.modal_fusion = nn.ModuleList([modal_fusion(in_channel=in_channel) for _ in range(num_cls)])
        self.region_fusion = region_fusion(in_channel=in_channel, num_cls=num_cls)
        self.short_cut = nn.Sequential(general_conv3d(in_channel * 4, in_channel, k_size=1, padding=0, stride=1),
                                       general_conv3d(in_channel, in_channel, k_size=3, padding=1, stride=1),
                                       general_conv3d(in_channel, in_channel // 2, k_size=1, padding=0, stride=1))
        self.clsname_list = ['BG', 'NCR', 'ED', 'NET', 'ET']

    def forward(self, x, prm, mask=None):
        ys = [f.t for f in x]
        msk = x[0].mask
        _, D, H, W, C = ys[0].shape
        B = prm.shape[0]                                          # virtual batch: P passes x physical samples of ys
        V = D * H * W
        S, P = ops.region_gap(prm, msk, ys)                       # (B,4,4,C) sums, (B,4) prob sums
        prm_avg = P / V + 1e-7                                    # blocks.py:507
        feat_avg = (S / V) / prm_avg.view(B, 4, 1, 1)             # blocks.py:508
        w = torch.stack([self.modal_fusion[i].weights(feat_avg[:, i].reshape(B, 4 * C).float(), prm_avg[:, i:i + 1])
                         for i in range(self.num_cls)], 1)        # (B, region, modality)
        R = ops.region_mix(prm, msk, w, ys)
        return [self.region_fusion(Feat(R)), _chain(self.short_cut, x)]


# ------------------------------------------------------------------------------------------------
# m2ftrans blocks: cross-token channel attention on the 8^3 / 16^3 token volumes
# ------------------------------------------------------------------------------------------------
def _bf16(t):
    return t.dtype == torch.bfloat16


def _pw(conv, x):
    """1x1x1 nn.Conv3d on a channels-last tensor = a token GEMM (hipBLASLt through ops.linear, bf16 operand images kept)"""
    return ops.linear(x, conv.weight, conv.bias, _bf16(x))


def _ln(norm, x, act=False, res=None, out_dtype=None):
    """act(LayerNorm over the channels of a channels-last tensor [+ res]), act = GELU or nothing: one fused launch
    (ops.ln_act, csrc/token_ops.hip: f32 statistics and affine, result in the tensor's dtype); channel counts the kernel does
    not take go through torch"""
    C = x.shape[-1]
    if x.is_cuda and C % 8 == 0 and C <= 512 and ((C // 8) & (C // 8 - 1)) == 0:
        return ops.ln_act(x, norm, act, res, out_dtype)
    y = torch.nn.functional.layer_norm(x.float(), (C,), norm.weight, norm.bias, norm.eps)
    y = y if x.dtype == torch.float32 else y.to(x.dtype)
    if res is not None:
        y = y + res
    return torch.nn.functional.gelu(y) if act else y


class DepthWiseConvBlock(nn.Module):
    """1x1 -> LN -> GELU -> depthwise 3x3x3 -> LN -> GELU -> 1x1 -> LN (reference blocks.py:32-63) on (B,D,H,W,C) tensors"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        mid = in_channels
        self.conv1 = nn.Conv3d(in_channels, mid, 1, 1)
        self.norm1 = nn.LayerNorm(mid, eps=1e-6)
        self.act1 = nn.GELU()
        self.conv2 = nn.Conv3d(mid, mid, 3, 1, 1, groups=mid)
        self.norm2 = nn.LayerNorm(mid, eps=1e-6)
        self.act2 = nn.GELU()
        self.conv3 = nn.Conv3d(mid, out_channels, 1, 1)
        self.norm3 = nn.LayerNorm(out_channels, eps=1e-6)

    def forward(self, x):
        x = _ln(self.norm1, _pw(self.conv1, x), act=True)
        x = _ln(self.norm2, ops.group_conv3d(x, self.conv2.weight, self.conv2.bias), act=True)
        return _ln(self.norm3, _pw(self.conv3, x))


def _ln_ok(x):
    C = x.shape[-1]
    return x.is_cuda and C % 8 == 0 and C <= 512 and ((C // 8) & (C // 8 - 1)) == 0


def dwcb_grouped(blocks, x):
    """G DepthWiseConvBlocks of equal shape, block g over x[g] ((G, B, D, H, W, C)), as ONE batch: grouped token GEMMs
    (ops.linear_grouped), LayerNorm + GELU with G affine sets (ops.ln_act), depthwise conv with G weight sets (ops.group_conv3d).
    The four key_map_* and four value_map_* blocks of MultiMaskAttentionLayer (reference blocks.py:246-297) cost 9 launches instead
    of 72 (and their backward 20 instead of ~200)."""
    G = len(blocks)
    _, B, D, H, W, C = x.shape
    if G == 1 or not _ln_ok(x):
        return torch.stack([b(x[g]) for g, b in enumerate(blocks)])
    R = B * D * H * W
    bf = _bf16(x)
    h = ops.linear_grouped(x.reshape(G, R, C), [b.conv1.weight for b in blocks], [b.conv1.bias for b in blocks], bf)
    h = ops.ln_act(h, [b.norm1 for b in blocks], True)
    mid = h.shape[-1]
    h = ops.group_conv3d(h.view(G * B, D, H, W, mid), [b.conv2.weight for b in blocks], [b.conv2.bias for b in blocks])
    h = ops.ln_act(h.view(G, R, mid), [b.norm2 for b in blocks], True)
    h = ops.linear_grouped(h, [b.conv3.weight for b in blocks], [b.conv3.bias for b in blocks], bf)
    return ops.ln_act(h, [b.norm3 for b in blocks], False).view(G, B, D, H, W, -1)


class GroupConvBlock(nn.Module):
    """reference blocks.py:65-109: 1x1 expand -> LN -> GELU -> grouped 3x3x3 (groups = embed_dims) -> LN -> GELU -> 1x1 -> LN,
    residual, GELU"""

    def __init__(self, embed_dims=8, expand_ratio=4, proj_drop=0.):
        super().__init__()
        self.pwconv1 = nn.Conv3d(embed_dims, embed_dims * expand_ratio, 1, 1)
        self.norm1 = nn.LayerNorm(embed_dims * expand_ratio, eps=1e-6)
        self.act1 = nn.GELU()
        self.dwconv = nn.Conv3d(embed_dims * expand_ratio, embed_dims * expand_ratio, 3, 1, 1, groups=embed_dims)
        self.norm2 = nn.LayerNorm(embed_dims * expand_ratio, eps=1e-6)
        self.act2 = nn.GELU()
        self.pwconv2 = nn.Conv3d(embed_dims * expand_ratio, embed_dims, 1, 1)
        self.norm3 = nn.LayerNorm(embed_dims, eps=1e-6)
        self.final_act = nn.GELU()
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, identity=None):
        h = _ln(self.norm1, _pw(self.pwconv1, x), act=True)
        h = _ln(self.norm2, ops.group_conv3d(h, self.dwconv.weight, self.dwconv.bias), act=True)
        idt = x if identity is None else identity
        if not (self.training and self.proj_drop.p > 0):
            return _ln(self.norm3, _pw(self.pwconv2, h), act=True, res=idt)        # LN -> + identity -> GELU in one launch
        h = _ln(self.norm3, _pw(self.pwconv2, h))
        return torch.nn.functional.gelu(idt + self.proj_drop(h))


class MultiMaskAttentionLayer(nn.Module):
    """channel attention of a query volume over the channels of the four modality volumes, absent modalities masked out
    (reference blocks.py:246-297, mask_gen_cross4 of mask.py:25-36)"""

    def __init__(self, kv_dim=8, query_dim=4, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.attn_drop = nn.Dropout(attn_drop)
        self.query_map = DepthWiseConvBlock(query_dim, query_dim)
        for m in ("flair", "t1ce", "t1", "t2"):
            setattr(self, f"key_map_{m}", DepthWiseConvBlock(kv_dim, kv_dim))
            setattr(self, f"value_map_{m}", DepthWiseConvBlock(kv_dim, kv_dim))
        self.out_project = DepthWiseConvBlock(query_dim, query_dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, query, key, value, mask):
        """query (B,D,H,W,Cq); key = value = the four modality volumes stacked (4,B,D,H,W,C); mask (B,4) bool"""
        B = query.shape[0]
        L_ = query.shape[1] * query.shape[2] * query.shape[3]
        mods = ("flair", "t1ce", "t1", "t2")
        q = self.query_map(query).reshape(B, L_, -1)
        blocks = [getattr(self, f"key_map_{m}") for m in mods] + [getattr(self, f"value_map_{m}") for m in mods]
        kv = dwcb_grouped(blocks, torch.cat([key, value]))                                                         # (8, B, D, H, W, C)
        C = kv.shape[-1]
        k = kv[:4].reshape(4, B, L_, C).permute(1, 2, 0, 3).reshape(B, L_, 4 * C)                                  # (B, L, 4C)
        v = kv[4:].reshape(4, B, L_, C).permute(1, 2, 0, 3).reshape(B, L_, 4 * C)
        att = torch.bmm(q.transpose(1, 2), k).float() * L_ ** -0.5                                                # (B, Cq, 4C)
        keep = mask.repeat_interleave(k.shape[2] // 4, dim=1).unsqueeze(1)
        att = self.attn_drop(att.masked_fill(~keep, float("-inf")).softmax(-1)).to(v.dtype)
        x = torch.bmm(v, att.transpose(1, 2)).reshape(query.shape)                                                 # (B, L, Cq)
        return query + self.proj_drop(self.out_project(x))


class MultiMaskCrossBlock(nn.Module):
    """reference blocks.py:206-244"""

    def __init__(self, feature_channels=128, num_classes=128, expand_ratio=4, drop_rate=0., attn_drop_rate=0., ffn_feature_maps=True):
        super().__init__()
        self.ffn_feature_maps = ffn_feature_maps
        self.cross_attn = MultiMaskAttentionLayer(kv_dim=feature_channels, query_dim=num_classes, attn_drop=attn_drop_rate,
                                                  proj_drop=drop_rate)
        if ffn_feature_maps:
            self.ffn2 = GroupConvBlock(embed_dims=feature_channels, expand_ratio=expand_ratio)
        self.ffn1 = GroupConvBlock(embed_dims=num_classes, expand_ratio=expand_ratio)

    def forward(self, kernels, feature_maps, mask):
        kernels = self.cross_attn(query=kernels, key=feature_maps, value=feature_maps, mask=mask)
        kernels = self.ffn1(kernels, identity=kernels)
        if self.ffn_feature_maps:                      # one GroupConvBlock (shared weights) over the four modality volumes: one batch
            fm = feature_maps.reshape((-1,) + tuple(feature_maps.shape[2:]))
            feature_maps = self.ffn2(fm, identity=fm).view(feature_maps.shape)
        return kernels, feature_maps


class fusion_postnorm(nn.Module):
    """masked concat of the four modality features -> 1x1 -> 3x3x3 -> 1x1 post-norm convs (reference blocks.py:567-580)"""

    def __init__(self, in_channel=64, num_cls=4):
        super().__init__()
        self.fusion_layer = nn.Sequential(general_conv3d(in_channel * num_cls, in_channel, k_size=1, padding=0, stride=1),
                                          general_conv3d(in_channel, in_channel, k_size=3, padding=1, stride=1),
                                          general_conv3d(in_channel, in_channel, k_size=1, padding=0, stride=1))

    def forward(self, x, mask=None):
        """x: the four modality Feats carrying (mask, column)"""
        return _chain(self.fusion_layer, x)


class prm_fusion_pk(nn.Module):
    """reference blocks.py:481-493 -> logits (channels-last f32)"""

    def __init__(self, in_channel=64, norm="in", num_cls=4):
        super().__init__()
        self.prm_layer = nn.Sequential(general_conv3d(in_channel, 16, k_size=1, stride=1, padding=0),
                                       nn.Conv3d(16, num_cls, kernel_size=1, padding=0, stride=1, bias=True))

    def forward(self, x1):
        return _chain(self.prm_layer, x1).t
