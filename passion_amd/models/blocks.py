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


class Feat:
    """A channels-last activation (B,D,H,W,C) plus its lazily computed InstanceNorm statistics and
    an optional modality mask (mask tensor, column): consumers treat absent modalities as zeros."""
    __slots__ = ("t", "_stats", "mask", "mask_col")

    def __init__(self, t, mask=None, mask_col=0, _stats=None):
        self.t, self.mask, self.mask_col = t, mask, mask_col
        self._stats = _stats if _stats is not None else [None]   # shared between masked views of one tensor

    def stats(self):
        if self._stats[0] is None:
            self._stats[0] = ops.instnorm_stats(self.t.detach())
        return self._stats[0]

    def with_mask(self, mask, col):
        return Feat(self.t, mask, col, self._stats)

    def seg(self, norm=True):
        return ops.SegIn(self.t, stats=self.stats() if norm else None, mask=self.mask, mask_col=self.mask_col)

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


def run_conv(conv, feats, norm, residual=None, out_dtype=None):
    """Launch the fused conv for an nn.Conv3d parameter holder over the virtual concat of `feats`."""
    if not isinstance(feats, (list, tuple)):
        feats = [feats]
    D, H, W = feats[0].dims
    cache = conv.__dict__.setdefault("_passion_pack", {})
    y = ops.conv_block([f.seg(norm) for f in feats], conv.weight, conv.bias, D, H, W,
                       ksize=conv.kernel_size[0], stride=conv.stride[0], pad_mode=PAD[conv.padding_mode],
                       residual=residual, out_dtype=out_dtype, pack_cache=cache)
    return Feat(y)


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

    def forward(self, feats, residual=None):
        return run_conv(self.conv, feats, norm=True, residual=residual)


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

    def forward(self, feats):
        h = self.fusion_layer[0](feats)
        h = self.fusion_layer[1](h)
        return self.fusion_layer[2](h)
