"""m2ftrans backbone + PASSION self-distillation losses on the MI355X HIP kernels.

Drop-in for the reference's code/models/m2ftrans.py (Model :473-771): same constructor, plain attributes (`mask_type`,
`use_passion`, `is_training`), forward signature / return tuples, module tree and state-dict keys (801 tensors), same
construction order (hence RNG stream).  `m2ftrans.patch_size = S // 16` selects the crop edge (the reference hard-codes 5 = 80^3).

Underneath: post-norm convs (general_conv3d = Conv -> IN -> LReLU) run on the fused conv kernels exactly like RFNet's (the
consumer's loader normalises + activates); the masked 5-stream bottleneck transformer runs its softmax(QK^T)V core on the
block-masked attention kernels (ops.masked_attention; 16-wide heads) and its token GEMMs / LayerNorm / GELU on torch (hipBLASLt,
SURVEY K8); Weight_Attention is one fused scale-by-nearest-upsampled-map kernel per level (ops.scale_nearest); the cross-token
blocks' depthwise / grouped 3x3x3 convs are ops.group_conv3d.  The teacher and the four uni-modal PASSION students (reference
:583-641) run as ONE batch of 5 B through the bottleneck and the fusion decoder.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from ..utils import criterions as crit
from .blocks import (GROUPED_ENC_POST as GROUPED_ENC, Feat, PFeat, MultiMaskCrossBlock, encoders_grouped_post, fusion_postnorm, general_conv3d, materialize, prm_fusion_pk, run_conv_auto)
from .mmformer import Linear

basic_dims = 8
transformer_basic_dims = 512
mlp_dim = 4096
num_heads = 8
depth = 3
num_cls = 4
num_modals = 4
patch_size = 5
HWD = 80
H = W = Z = 80
MODALS = ("flair", "t1ce", "t1", "t2")


def _up2_mat(feats):
    """trilinear x2 of the activated value of one Feat / of the channel concat of several"""
    if not isinstance(feats, (list, tuple)):
        feats = [feats]
    mats = [materialize(f) for f in feats]
    t = mats[0].t if len(mats) == 1 else torch.cat([m.t for m in mats], -1)
    return Feat(ops.upsample(t, 2), feats[0].mask, feats[0].mask_col)


class MultiCrossToken(nn.Module):
    """two MultiMaskCrossBlocks (reference m2ftrans.py:25-61)"""

    def __init__(self, image_h=80, image_w=80, image_d=80, h_stride=16, w_stride=16, d_stride=16, num_layers=2, mlp_ratio=4,
                 drop_rate=0.1, attn_drop_rate=0.0, interpolate_mode="trilinear", channel=basic_dims * 16):
        super().__init__()
        self.channels = channel
        self.layers = nn.ModuleList([MultiMaskCrossBlock(feature_channels=channel, num_classes=channel, expand_ratio=mlp_ratio,
                                                         drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
                                                         ffn_feature_maps=i != num_layers - 1) for i in range(num_layers)])

    def forward(self, inputs, kernels, mask):
        feature_maps = torch.stack(list(inputs))        # (4, B, D, H, W, C): the blocks treat the four modalities as one batch
        for layer in self.layers:
            kernels, feature_maps = layer(kernels, feature_maps, mask)
        return kernels


class Encoder(nn.Module):
    """five-level post-norm residual encoder of one modality (reference m2ftrans.py:63-104)"""

    def __init__(self):
        super().__init__()
        c = basic_dims
        self.e1_c1 = general_conv3d(1, c, pad_type="reflect")
        self.e1_c2 = general_conv3d(c, c, pad_type="reflect")
        self.e1_c3 = general_conv3d(c, c, pad_type="reflect")
        for lvl in range(2, 6):
            ci, co = c * 2 ** (lvl - 2), c * 2 ** (lvl - 1)
            setattr(self, f"e{lvl}_c1", general_conv3d(ci, co, stride=2, pad_type="reflect"))
            setattr(self, f"e{lvl}_c2", general_conv3d(co, co, pad_type="reflect"))
            setattr(self, f"e{lvl}_c3", general_conv3d(co, co, pad_type="reflect"))

    def forward(self, x, modal, mask, dtype):
        B, K, D, Hh, Ww = x.shape
        vol = D * Hh * Ww
        seg = ops.SegIn(x, mask=mask, mask_col=modal, C=1, offset=modal * vol, bstride=K * vol)
        conv = self.e1_c1.conv
        cache = conv.__dict__.setdefault("_passion_pack", {})
        r1 = PFeat(ops.conv_block([seg], conv.weight, conv.bias, D, Hh, Ww, ksize=3, stride=1, pad_mode=L.PAD_REFLECT,
                                  out_dtype=dtype, pack_cache=cache))
        feats = []
        for lvl in range(1, 6):
            if lvl > 1:
                r1 = getattr(self, f"e{lvl}_c1")(feats[-1])
            r3 = getattr(self, f"e{lvl}_c3")(getattr(self, f"e{lvl}_c2")(r1))
            feats.append(Feat(ops.norm_act(r1.t, r1.stats(), r3.t, r3.stats())))      # x + c3(c2(x)) on activated values
        return feats


class Decoder_sep(nn.Module):
    """reference m2ftrans.py:106-154 -> softmax probabilities (channels-last f32)"""

    def __init__(self, num_cls=4):
        super().__init__()
        c = basic_dims
        for lvl in (4, 3, 2, 1):
            co = c * 2 ** (lvl - 1)
            setattr(self, f"d{lvl}", nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True))
            setattr(self, f"d{lvl}_c1", general_conv3d(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_c2", general_conv3d(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_out", general_conv3d(co, co, k_size=1, padding=0, pad_type="reflect"))
        self.seg_layer = nn.Conv3d(in_channels=c, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x1, x2, x3, x4, x5, probs=True):
        skips = (x1, x2, x3, x4)
        h = self.d4_c1(_up2_mat(x5))
        for lvl in (4, 3, 2, 1):
            h = getattr(self, f"d{lvl}_out")(getattr(self, f"d{lvl}_c2")([h, skips[lvl - 1]]))
            if lvl > 1:
                h = getattr(self, f"d{lvl - 1}_c1")(_up2_mat(h))
        logits = run_conv_auto(self.seg_layer, h, out_dtype=torch.float32).t
        return ops.softmax_classes(logits, 1.0, "cl") if probs else logits      # (logits: ops.sep_wce_dice forms the softmax itself)


class Decoder_fusion(nn.Module):
    """reference m2ftrans.py:157-254"""

    def __init__(self, num_cls=4):
        super().__init__()
        c = basic_dims
        self.d5_c2 = general_conv3d(c * 32, c * 16, pad_type="reflect")
        self.d5_out = general_conv3d(c * 16, c * 16, k_size=1, padding=0, pad_type="reflect")
        self.CT5 = MultiCrossToken(h_stride=16, w_stride=16, d_stride=16, channel=c * 16)
        self.CT4 = MultiCrossToken(h_stride=8, w_stride=8, d_stride=8, channel=c * 8)
        for lvl in (4, 3, 2, 1):
            co = c * 2 ** (lvl - 1)
            setattr(self, f"d{lvl}_c1", general_conv3d(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_c2", general_conv3d(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_out", general_conv3d(co, co, k_size=1, padding=0, pad_type="reflect"))
        self.seg_layer = nn.Conv3d(in_channels=c, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.up2 = nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
        self.RFM3 = fusion_postnorm(in_channel=c * 4, num_cls=num_cls)
        self.RFM2 = fusion_postnorm(in_channel=c * 2, num_cls=num_cls)
        self.RFM1 = fusion_postnorm(in_channel=c * 1, num_cls=num_cls)
        self.prm_fusion5 = prm_fusion_pk(in_channel=c * 16, num_cls=num_cls)
        self.prm_fusion4 = prm_fusion_pk(in_channel=c * 8, num_cls=num_cls)
        self.prm_fusion3 = prm_fusion_pk(in_channel=c * 4, num_cls=num_cls)
        self.prm_fusion2 = prm_fusion_pk(in_channel=c * 2, num_cls=num_cls)
        self.prm_fusion1 = prm_fusion_pk(in_channel=c * 1, num_cls=num_cls)

    def forward(self, dx1, dx2, dx3, dx4, dx5, fusion, mask):
        """dx1..dx3: lists of the four weighted modality Feats (carrying the pass mask); dx4, dx5: tuples of four plain tensors;
        fusion: (B,p,p,p,C) token volume; mask (B,4) bool.  -> logits, five deep-supervision logits, the reference's (de_x1..de_x5)"""
        preds = {5: self.prm_fusion5(Feat(fusion))}
        ct = self.CT5(dx5, fusion, mask)
        h = self.d5_out(self.d5_c2([Feat(ct), Feat(fusion)]))
        h = self.d4_c1(_up2_mat(h))
        feats = {5: h}
        preds[4] = self.prm_fusion4(h)
        hm = materialize(h)
        ct = self.CT4(dx4, hm.t, mask)
        g = self.d4_out(self.d4_c2([Feat(ct), h]))
        h = self.d3_c1(_up2_mat(g))
        feats[4] = h
        for lvl, dxl in ((3, dx3), (2, dx2), (1, dx1)):
            preds[lvl] = getattr(self, f"prm_fusion{lvl}")(h)
            g = getattr(self, f"d{lvl}_out")(getattr(self, f"d{lvl}_c2")([getattr(self, f"RFM{lvl}")(dxl), h]))
            if lvl > 1:
                h = getattr(self, f"d{lvl - 1}_c1")(_up2_mat(g))
                feats[lvl] = h
            else:
                feats[1] = g
        logits = run_conv_auto(self.seg_layer, feats[1], out_dtype=torch.float32).t
        return logits, tuple(preds[i] for i in (1, 2, 3, 4, 5)), tuple(feats[i] for i in (1, 2, 3, 4, 5))


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x) + x


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x):
        return self.fn(self.norm(x))


class GELU(nn.Module):
    def forward(self, x):
        return F.gelu(x)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout_rate):
        super().__init__()
        self.net = nn.Sequential(Linear(dim, hidden_dim), GELU(), nn.Dropout(p=dropout_rate), Linear(hidden_dim, dim),
                                 nn.Dropout(p=dropout_rate))

    def forward(self, x):
        return self.net(x)


class MaskedResidual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, mask):
        y, attn = self.fn(x, mask)
        return y + x, attn


class MaskedPreNormDrop(nn.Module):
    def __init__(self, dim, dropout_rate, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.fn = fn

    def forward(self, x, mask):
        x, attn = self.fn(self.norm(x), mask)
        return self.dropout(x), attn


class MaskedAttention(nn.Module):
    """reference m2ftrans.py:318-351; `allow`: the (B,5,5) block mask of mask_gen_fusion (mask.py:5-22) as bytes"""

    def __init__(self, dim, heads=8, qkv_bias=False, qk_scale=None, dropout_rate=0.0, num_class=4):
        super().__init__()
        self.num_heads = heads
        head_dim = dim // heads
        self.scale = qk_scale or head_dim ** -0.5
        self.num_class = num_class
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(dropout_rate)
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(dropout_rate)

    def forward(self, x, allow):
        qkv = self.qkv(x)
        p = self.attn_drop.p if self.training else 0.0
        out, attn = ops.masked_attention(qkv, allow, self.num_heads, self.scale, p)
        return self.proj_drop(self.proj(out)), attn


class MaskedTransformer(nn.Module):
    """reference m2ftrans.py:354-391"""

    def __init__(self, embedding_dim, depth, heads, mlp_dim, dropout_rate=0.1, n_levels=1, n_points=4):
        super().__init__()
        att, ffn = [], []
        self.depth = depth
        for _ in range(depth):
            att.append(MaskedResidual(MaskedPreNormDrop(embedding_dim, dropout_rate,
                                                        MaskedAttention(embedding_dim, heads=heads, dropout_rate=dropout_rate))))
            ffn.append(Residual(PreNorm(embedding_dim, FeedForward(embedding_dim, mlp_dim, dropout_rate))))
        self.cross_attention_list = nn.ModuleList(att)
        self.cross_ffn_list = nn.ModuleList(ffn)

    def forward(self, x, allow):
        attn_list = []
        for j in range(self.depth):
            x, attn = self.cross_attention_list[j](x, allow)
            attn_list.append(attn.detach())
            x = self.cross_ffn_list[j](x)
        return x, attn_list


def fusion_allow(mask):
    """mask_gen_fusion (reference mask.py:5-22) in block form: (B,5,5) bytes -- a modality's tokens attend to themselves, the
    fusion tokens to the present modalities and to themselves"""
    B = mask.shape[0]
    allow = torch.eye(5, dtype=torch.uint8, device=mask.device).unsqueeze(0).repeat(B, 1, 1)      # (no index tensors: capturable)
    allow[:, 4, :4] = mask.to(torch.uint8)
    return allow


class Bottleneck(nn.Module):
    """reference m2ftrans.py:393-411; tokens (B, 5 n, C) in, the five token sets + the attention probabilities out"""

    def __init__(self):
        super().__init__()
        self.trans_bottle = MaskedTransformer(embedding_dim=basic_dims * 16, depth=depth, heads=num_heads, mlp_dim=mlp_dim)
        self.num_cls = num_modals

    def forward(self, embed_cat, mask, bf16):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            tok, attn = self.trans_bottle(embed_cat, fusion_allow(mask))
        return torch.chunk(tok, self.num_cls + 1, dim=1), attn


class Weight_Attention(nn.Module):
    """reference m2ftrans.py:413-460: kept for tree fidelity; Model.forward runs it as ops.scale_nearest launches"""

    def __init__(self):
        super().__init__()
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")


class MaskModal(nn.Module):
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
        self.Bottleneck = Bottleneck()
        self.decoder_fusion = Decoder_fusion(num_cls=num_cls)
        self.decoder_sep = Decoder_sep(num_cls=num_cls)
        self.weight_attention = Weight_Attention()
        self.masker = MaskModal()
        n = patch_size ** 3
        self.pos = nn.Parameter(torch.zeros(1, n * 5, basic_dims * 16))
        self.fusion = nn.Parameter(nn.init.normal_(torch.zeros(1, n, basic_dims * 16), mean=0.0, std=1.0))
        eye = np.eye(4, dtype=bool)
        for i in range(4):
            setattr(self, f"masks_mod{i}", torch.from_numpy(eye[i:i + 1].copy()))
        self.up1 = nn.Identity()
        self.up2 = nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
        self.up4 = nn.Upsample(scale_factor=4, mode="trilinear", align_corners=True)
        self.up8 = nn.Upsample(scale_factor=8, mode="trilinear", align_corners=True)
        self.up16 = nn.Upsample(scale_factor=16, mode="trilinear", align_corners=True)
        self.up_ops = nn.ModuleList([self.up1, self.up2, self.up4, self.up8, self.up16])
        self.mask_type = "idt"
        self.is_training = False
        self.use_passion = False
        self.compute_dtype = torch.float32
        self.num_cls = num_cls
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                torch.nn.init.kaiming_normal_(m.weight)

    def _onehots_on(self, dev):
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
        ops.repack_stale()
        ops.recast_stale()
        B = x.size(0)
        dev, dt = x.device, self.compute_dtype
        bf16 = dt == torch.bfloat16
        idt = self.mask_type != "pdt"
        training = self.is_training
        passion = training and self.use_passion
        mask = mask.to(device=dev, dtype=torch.bool).contiguous()
        xin = x.to(dt).contiguous()
        grouped = GROUPED_ENC and bf16
        if grouped:
            stk = encoders_grouped_post([getattr(self, f"{m}_encoder") for m in MODALS], xin, mask if idt else None, dt, 5)
            parts = [ops.unstack(stk[l].alt if stk[l].alt is not None else stk[l].t, 4) for l in range(5)]      # storage-sharing slices; backward: one cat, no zero-filled slices
            encs = [[Feat(parts[l][i]) for l in range(5)] for i in range(4)]
        else:
            encs = [getattr(self, f"{m}_encoder")(xin, i, mask if idt else None, dt) for i, m in enumerate(MODALS)]
        def gated(t, i):
            return t * mask[:, i].view(B, 1, 1, 1, 1).to(t.dtype) if idt else t

        # ---- the masks of every pass through the bottleneck + fusion decoder: teacher (data mask) + four uni-modal students
        # (pure one-hots, as the reference passes masks_mod0..3 at :585-641), as ONE batch of P * B, pass-major
        P, masks_all = 1, mask
        if passion:
            P = 5
            onehots = [t.expand(B, 4) for t in self._onehots_on(dev)]
            masks_all = torch.cat([mask] + onehots, 0).contiguous()
        p = encs[0][4].t.shape[1]
        n = p ** 3
        C5 = encs[0][4].t.shape[-1]
        toks = [gated(encs[i][4].t, i).reshape(B, n, C5) for i in range(4)]
        embed = torch.cat(toks + [self.fusion.expand(B, -1, -1).to(toks[0].dtype)], 1).float() + self.pos       # f32 residual stream
        if P > 1:
            embed = embed.repeat(P, 1, 1)
        chunks, attn = self.Bottleneck(embed, masks_all, bf16)
        vols = [c.to(dt).reshape(P * B, p, p, p, C5).contiguous() for c in chunks]      # chunk() views keep the (5 n C) batch stride
        # Weight_Attention (:420-460): first layer, fusion queries, summed over heads and queries -> one (p,p,p) map per modality
        wsum = attn[0][:, :, 4 * n:, :].sum((1, 2))                                   # (PB, 5n)
        wmaps = [wsum[:, i * n:(i + 1) * n].reshape(P * B, p, p, p) for i in range(4)]
        dx5 = tuple(vols[i] * wmaps[i].unsqueeze(-1).to(dt) for i in range(4))
        # (the encoder features of a modality that is absent in the data are exactly zero already: zero input -> constant conv
        # output -> InstanceNorm of a constant = 0 at every level, as the reference's explicit masking at :529-539 leaves them)
        dxs = {}
        for lvl in (4, 3, 2, 1):
            dxs[lvl] = [ops.scale_nearest(encs[i][lvl - 1].t, wmaps[i], 5 - lvl) for i in range(4)]
        as_feats = lambda ts: [Feat(t, masks_all, i) for i, t in enumerate(ts)]        # noqa: E731
        fuse_all, preds_all, de_f_all = self.decoder_fusion(as_feats(dxs[1]), as_feats(dxs[2]), as_feats(dxs[3]), tuple(dxs[4]), dx5,
                                                            vols[4], masks_all)
        if P == 1:
            fuse_pred, preds = fuse_all, preds_all
        else:
            fuse_pred, fuse_stud = fuse_all.split([B, 4 * B], 0)
            preds_ts = [pp.split([B, 4 * B], 0) for pp in preds_all]
            preds, preds_stud = [t[0] for t in preds_ts], [t[1] for t in preds_ts]
        pred = ops.softmax_classes(fuse_pred, 1.0, "ncdhw")
        if not training:
            return pred

        tcl = ops.target_to_cl(target)
        sep_mask = mask.t().reshape(4 * B, 1).contiguous() if idt else None           # row i*B+b = mask[b, i]
        if grouped:
            stacked = [Feat(stk[l].t, sep_mask, 0) for l in range(5)]
        else:
            stacked = [Feat(torch.cat([encs[i][l].t for i in range(4)], 0), sep_mask, 0) for l in range(5)]
        sep_all = self.decoder_sep(*stacked, probs=not ops.SEP_LOSS_FUSED)
        eye = torch.eye(4, device=dev)
        maskf = mask.float()
        gates = [(maskf * eye[i]) if idt else eye[i].expand(B, 4) for i in range(4)]
        ups = (1, 2, 4, 8, 16)

        prm_loss = torch.zeros(B, 1, device=dev)
        for j, pp in enumerate(preds):                                                # weights 1/2 .. 1/32 (:649-653)
            pair = crit.seg_losses_cl(ops.softmax_classes(pp, 1.0, "cl"), tcl, ups[j])
            prm_loss = prm_loss + 0.5 ** (j + 1) * pair.sum(1, keepdim=True)
        sep_pair = ops.sep_wce_dice(sep_all, tcl).sum(1, keepdim=True).view(4, B, 1)
        sep_loss = torch.zeros(B, 4, device=dev)
        for i in range(4):
            sep_loss = sep_loss + gates[i] * sep_pair[i]
        if not passion:
            return pred, prm_loss, sep_loss

        inv_t = 1.0 / temp
        pt_full = ops.softmax_classes(fuse_pred.detach(), inv_t, "cl")
        pt_deep = [ops.softmax_classes(pp.detach(), inv_t, "cl") for pp in preds]        # coarse: crit.kl_cl upsamples inside the kernel
        f1_teacher, f1_stud = materialize(de_f_all[0]).t.split([B, 4 * B], 0)
        pl_all, d_all = ops.proto_loss(f1_stud, f1_teacher.detach(), tcl)
        kl_all = crit.kl_cl(fuse_stud, pt_full, temp)
        for j in range(5):
            kl_all = kl_all + 0.5 ** (j + 1) * crit.kl_cl(preds_stud[j], pt_deep[j], temp, ups[j])
        pl_all, d_all, kl_all = pl_all.view(4, B, 1), d_all.view(4, B, 1), kl_all.view(4, B, 1)
        kl_loss, proto_loss, dist = (torch.zeros(B, 4, device=dev) for _ in range(3))
        for i in range(4):
            # a sample whose data mask IS one-hot i feeds student i exactly what the teacher saw: the reference gets 0 exactly
            # (and train.py:266-267 turns that 0/0 into an all-false rp_mask); rows of one batched launch may round differently
            # -- with dropout OFF.  Under model.train() the reference runs every student through its own Bottleneck / decoder_fusion
            # call with fresh dropout (m2ftrans.py:585-641: MaskedTransformer 0.1, MultiCrossToken 0.1): student != teacher, the terms
            # are positive and feed epoch_dist_m / the relative-preference gate; the rows of the batched launch draw different
            # dropout masks here as well, so nothing is imposed then
            differs = (~(mask == onehots[i]).all(1)).float().view(B, 1)
            if self.training:
                differs = torch.ones_like(differs)
            proto_loss = proto_loss + gates[i] * pl_all[i] * differs
            dist = dist + gates[i] * d_all[i] * differs
            kl_loss = kl_loss + gates[i] * kl_all[i] * differs
        return pred, prm_loss, sep_loss, kl_loss, proto_loss, dist
