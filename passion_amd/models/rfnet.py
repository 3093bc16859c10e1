"""RFNet backbone + PASSION self-distillation losses on the MI355X HIP kernels.

Drop-in for the reference's code/models/rfnet.py (Model :176-403): same constructor, attributes
(`mask_type`, `use_passion`, `is_training`), forward signature / return tuples and state-dict keys.
Post-norm convs (general_conv3d = Conv -> IN -> LReLU) run on the same fused conv kernel as mmformer's
pre-norm blocks: a conv returns its raw output, the consumer's loader normalises + activates it; the
activated tensor is only materialised (ops.norm_act) for the encoder residuals, upsampling inputs and
prototype features.  Region-aware modal fusion runs on region_gap / region_mix (blocks.py).
"""
import os
import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..utils import criterions as crit
from .blocks import (GROUPED_ENC_POST, Feat, PFeat, encoders_grouped_post, general_conv3d, materialize, prm_generator_laststage_pk, prm_generator_pk,
                     region_aware_modal_fusion, run_conv_auto)

SIDE_BRANCHES = os.environ.get("PASSION_RFNET_SIDE", "1") != "0"      # decoder_sep and the deep-supervision losses on side streams (0: all on one stream)
basic_dims = 8
H = W = Z = 80
num_cls = 4
num_modals = 4
MODALS = ("flair", "t1ce", "t1", "t2")


def _up2_mat(feats):
    """upsample x2 of the activated value of one Feat / the channel concat of several"""
    if not isinstance(feats, (list, tuple)):
        feats = [feats]
    mats = [materialize(f) for f in feats]
    t = mats[0].t if len(mats) == 1 else torch.cat([m.t for m in mats], -1)
    return Feat(ops.upsample(t, 2), feats[0].mask, feats[0].mask_col)


class Encoder(nn.Module):
    """four-level post-norm residual encoder (reference rfnet.py:15-48)"""

    def __init__(self):
        super().__init__()
        c = basic_dims
        self.e1_c1 = general_conv3d(1, c, pad_type="reflect")
        self.e1_c2 = general_conv3d(c, c, pad_type="reflect")
        self.e1_c3 = general_conv3d(c, c, pad_type="reflect")
        for lvl in range(2, 5):
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
        for lvl in range(1, 5):
            if lvl > 1:
                r1 = getattr(self, f"e{lvl}_c1")(feats[-1])
            r3 = getattr(self, f"e{lvl}_c3")(getattr(self, f"e{lvl}_c2")(r1))
            feats.append(Feat(ops.norm_act(r1.t, r1.stats(), r3.t, r3.stats())))      # x + c3(c2(x)) on activated values
        return feats


class Decoder_sep(nn.Module):
    """reference rfnet.py:50-89 -> softmax probabilities (channels-last f32)"""

    def __init__(self, num_cls=4):
        super().__init__()
        c = basic_dims
        for lvl in (3, 2, 1):
            co = c * 2 ** (lvl - 1)
            setattr(self, f"d{lvl}", nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True))
            setattr(self, f"d{lvl}_c1", general_conv3d(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_c2", general_conv3d(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_out", general_conv3d(co, co, k_size=1, padding=0, pad_type="reflect"))
        self.seg_layer = nn.Conv3d(in_channels=c, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x1, x2, x3, x4, probs=True):
        skips = (x1, x2, x3)
        h = self.d3_c1(_up2_mat(x4))
        for lvl in (3, 2, 1):
            h = getattr(self, f"d{lvl}_out")(getattr(self, f"d{lvl}_c2")([h, skips[lvl - 1]]))
            if lvl > 1:
                h = getattr(self, f"d{lvl - 1}_c1")(_up2_mat(h))
        logits = run_conv_auto(self.seg_layer, h, out_dtype=torch.float32).t
        return ops.softmax_classes(logits, 1.0, "cl") if probs else logits      # (logits: ops.sep_wce_dice forms the softmax itself)


class Decoder_fuse(nn.Module):
    """reference rfnet.py:91-152"""

    def __init__(self, num_cls=4):
        super().__init__()
        c = basic_dims
        for lvl in (3, 2, 1):
            co = c * 2 ** (lvl - 1)
            setattr(self, f"d{lvl}_c1", general_conv3d(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_c2", general_conv3d(co * 2, co, pad_type="reflect"))
            setattr(self, f"d{lvl}_out", general_conv3d(co, co, k_size=1, padding=0, pad_type="reflect"))
        self.seg_layer = nn.Conv3d(in_channels=c, out_channels=num_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.softmax = nn.Softmax(dim=1)
        self.up2 = nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
        self.up4 = nn.Upsample(scale_factor=4, mode="trilinear", align_corners=True)
        self.up8 = nn.Upsample(scale_factor=8, mode="trilinear", align_corners=True)
        self.RFM4 = region_aware_modal_fusion(in_channel=c * 8, num_cls=num_cls)
        self.RFM3 = region_aware_modal_fusion(in_channel=c * 4, num_cls=num_cls)
        self.RFM2 = region_aware_modal_fusion(in_channel=c * 2, num_cls=num_cls)
        self.RFM1 = region_aware_modal_fusion(in_channel=c * 1, num_cls=num_cls)
        self.prm_generator4 = prm_generator_laststage_pk(in_channel=c * 8, num_cls=num_cls)
        self.prm_generator3 = prm_generator_pk(in_channel=c * 4, num_cls=num_cls)
        self.prm_generator2 = prm_generator_pk(in_channel=c * 2, num_cls=num_cls)
        self.prm_generator1 = prm_generator_pk(in_channel=c * 1, num_cls=num_cls)

    def forward(self, x1, x2, x3, x4, mask=None):
        """x1..x4: lists of the four modality Feats carrying the effective mask"""
        xs = {1: x1, 2: x2, 3: x3, 4: x4}
        prm4 = self.prm_generator4(x4)
        prms = {4: prm4}
        fused = self.RFM4(x4, ops.softmax_classes(prm4.detach(), 1.0, "cl"))
        h = self.d3_c1(_up2_mat(fused))                                        # de_x4
        feats = {4: h}
        for lvl in (3, 2, 1):
            prm = getattr(self, f"prm_generator{lvl}")(h, xs[lvl])
            prms[lvl] = prm
            fused = getattr(self, f"RFM{lvl}")(xs[lvl], ops.softmax_classes(prm.detach(), 1.0, "cl"))
            h = getattr(self, f"d{lvl}_out")(getattr(self, f"d{lvl}_c2")(fused + [h]))
            feats[lvl] = h
            if lvl > 1:
                h = getattr(self, f"d{lvl - 1}_c1")(_up2_mat(h))
                feats[lvl] = feats[lvl]
                nxt = h
            # note: reference returns de_x{lvl} AFTER d{lvl-1}_c1(up2(.)) for lvl 3,2 (rfnet.py:135,141) -- keep that
            if lvl > 1:
                feats[lvl] = nxt
        logits = run_conv_auto(self.seg_layer, feats[1], out_dtype=torch.float32).t
        return logits, (prms[1], prms[2], prms[3], prms[4]), (feats[1], feats[2], feats[3], feats[4])


class MaskModal(nn.Module):
    def forward(self, x, mask):
        B, K, C, Hh, Ww, Zz = x.size()
        return (x * mask.view(B, K, 1, 1, 1, 1).to(x.dtype)).view(B, -1, Hh, Ww, Zz)


class MaskModal_NoCat(nn.Module):
    def forward(self, x, mask):
        B, K = x.shape[:2]
        return x * mask.view(B, K, 1, 1, 1, 1).to(x.dtype)


class Model(nn.Module):
    def __init__(self, num_cls=4):
        super().__init__()
        self.flair_encoder = Encoder()
        self.t1ce_encoder = Encoder()
        self.t1_encoder = Encoder()
        self.t2_encoder = Encoder()
        self.decoder_fuse = Decoder_fuse(num_cls=num_cls)
        self.decoder_sep = Decoder_sep(num_cls=num_cls)
        self.masker = MaskModal()
        self.masker_nocat = MaskModal_NoCat()
        self.is_training = False
        self.use_passion = False
        self.mask_type = "idt"
        self.compute_dtype = torch.float32
        eye = np.eye(4, dtype=bool)
        for i in range(4):
            setattr(self, f"masks_mod{i}", torch.from_numpy(eye[i:i + 1].copy()))
        self.up1 = nn.Identity()
        self.up2 = nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
        self.up4 = nn.Upsample(scale_factor=4, mode="trilinear", align_corners=True)
        self.up8 = nn.Upsample(scale_factor=8, mode="trilinear", align_corners=True)
        self.up_ops = nn.ModuleList([self.up1, self.up2, self.up4, self.up8])
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                torch.nn.init.kaiming_normal_(m.weight)

    def _enc_streams(self, dev):
        cache = self.__dict__.setdefault("_streams_dev", {})
        if dev not in cache:
            n = max(1, min(4, int(os.environ.get("PASSION_SIDE_STREAMS", "4"))))      # < 4: the four roles share n streams (A/B)
            pool = [torch.cuda.Stream(device=dev) for _ in range(n)]
            cache[dev] = [pool[i % n] for i in range(4)]
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
        B = x.size(0)
        dev, dt = x.device, self.compute_dtype
        idt = self.mask_type != "pdt"
        mask = mask.to(device=dev, dtype=torch.bool).contiguous()
        xin = x.to(dt).contiguous()
        # the four encoders are independent: one HIP stream each (forward and, through autograd, backward), see mmformer.py
        from .mmformer import ENC_STREAMS
        streams = self._enc_streams(dev) if ENC_STREAMS else None
        main = torch.cuda.current_stream(dev) if streams else None
        encs = [None] * 4
        grouped = GROUPED_ENC_POST and dt == torch.bfloat16
        if grouped:          # every encoder layer as one grouped launch over the 4B stacked samples
            stk = encoders_grouped_post([getattr(self, f"{m}_encoder") for m in MODALS], xin, mask if idt else None, dt, 4)
            parts = [ops.unstack(stk[l].alt if stk[l].alt is not None else stk[l].t, 4) for l in range(4)]      # storage-sharing slices; backward: one cat, no zero-filled slices
            encs = [[Feat(parts[l][i]) for l in range(4)] for i in range(4)]
        for i, m in enumerate(MODALS):
            if grouped:
                break
            if streams:
                streams[i].wait_stream(main)
                with torch.cuda.stream(streams[i]):
                    encs[i] = getattr(self, f"{m}_encoder")(xin, i, mask if idt else None, dt)
                    for f in encs[i]:
                        f.t.record_stream(main)
                        if f._stats[0] is not None:
                            f._stats[0].record_stream(main)
            else:
                encs[i] = getattr(self, f"{m}_encoder")(xin, i, mask if idt else None, dt)
        if streams and not grouped:
            for s_ in streams:
                main.wait_stream(s_)
        training = self.is_training
        passion = training and self.use_passion

        # decoder_fuse always masks with the mask it is given (rfnet.py:126-144); in idt the features are masked too.
        # PASSION: teacher + four uni-modal students (rfnet.py:269-275) share the encoder features -> ONE virtual batch
        # of 5*B through the fusion path (conv segments, region_gap / region_mix re-read sample b % B under mask row b).
        P, masks_all = 1, mask
        if passion:
            P = 5
            onehots = [t.expand(B, 4).contiguous() for t in self._onehots_on(dev)]
            effs = [(mask & onehots[i]) if idt else onehots[i] for i in range(4)]
            masks_all = torch.cat([mask] + effs, 0).contiguous()
        eye = torch.eye(4, device=dev)
        maskf = mask.float()
        gates = [(maskf * eye[i]) if idt else eye[i].expand(B, 4) for i in range(4)]
        ups = (1, 2, 4, 8)
        tcl = ops.target_to_cl(target) if training else None

        def sep_branch():
            """decoder_sep (shared weights) once over the four modalities stacked along the batch axis + its loss: depends on the encoders
            only, like the fusion path"""
            sep_mask = mask.t().reshape(4 * B, 1).contiguous() if idt else None       # row i*B+b = mask[b, i]
            if grouped:
                stacked = [Feat(stk[l].t, sep_mask, 0) for l in range(4)]
            else:
                stacked = [Feat(torch.cat([encs[i][l].t for i in range(4)], 0), sep_mask, 0) for l in range(4)]
            sep_all = self.decoder_sep(*stacked, probs=not ops.SEP_LOSS_FUSED)
            sep_pair = ops.sep_wce_dice(sep_all, tcl).sum(1, keepdim=True).view(4, B, 1)      # 4B predictions against the B label volumes
            sep_loss = torch.zeros(B, 4, device=dev)
            for i in range(4):
                sep_loss = sep_loss + gates[i] * sep_pair[i]
            return sep_loss

        # Side streams (as models/mmformer.py; the encoders run grouped on the main stream here): decoder_sep with its loss on one, under
        # the fusion decoder -- forward, and (autograd replays a node on its forward's stream) backward; the deep-supervision losses
        # on another.  Until round 5 every launch of this model sat on ONE stream (1233 dispatches, 99.6 % of the step with exactly
        # one kernel in flight): the deep-level launches of the two decoders now fill each other's gaps.
        side = streams if (streams and training and SIDE_BRANCHES) else None
        sep_loss = None
        if side:
            s0 = side[0]
            s0.wait_stream(main)
            for t_ in [tcl, mask] + gates + ([f.t for f in stk] if grouped else [f.t for e in encs for f in e]):
                t_.record_stream(s0)                      # made on the main stream, read (forward and backward) on this one
            with torch.cuda.stream(s0):
                sep_loss = sep_branch()
                sep_loss.record_stream(main)
        fuse_all, preds_all, de_f_all = self.decoder_fuse(*[[encs[i][l].with_mask(masks_all, i, P) for i in range(4)]
                                                            for l in range(4)])
        if P == 1:
            fuse_pred, preds = fuse_all, preds_all
        else:
            # rows [0, B): the teacher; rows [B, 5B): the four uni-modal students, student-major
            fuse_pred, fuse_stud = fuse_all.split([B, 4 * B], 0)
            preds_ts = [pp.split([B, 4 * B], 0) for pp in preds_all]
            preds, preds_stud = [t[0] for t in preds_ts], [t[1] for t in preds_ts]
        pred = ops.softmax_classes(fuse_pred, 1.0, "ncdhw")
        if not training:
            return pred

        def deep_supervision():
            prm = torch.zeros(B, 1, device=dev)
            for j, pp in enumerate(preds):
                pair = crit.seg_losses_cl(ops.softmax_classes(pp, 1.0, "cl"), tcl, ups[j])
                prm = prm + 0.5 ** (j + 1) * pair.sum(1, keepdim=True)
            return prm

        if side:
            s2 = side[2]
            s2.wait_stream(main)
            tcl.record_stream(s2)
            for pp in preds:
                pp.record_stream(s2)
            with torch.cuda.stream(s2):
                prm_loss = deep_supervision()
                prm_loss.record_stream(main)
            main.wait_stream(s0)
            main.wait_stream(s2)
        else:
            sep_loss = sep_branch()
            prm_loss = deep_supervision()
        if not passion:
            return pred, prm_loss, sep_loss

        inv_t = 1.0 / temp
        pt_full = ops.softmax_classes(fuse_pred.detach(), inv_t, "cl")
        pt_deep = [ops.softmax_classes(pp.detach(), inv_t, "cl") for pp in preds]        # coarse: crit.kl_cl upsamples inside the kernel
        f1_teacher, f1_stud = materialize(de_f_all[0]).t.split([B, 4 * B], 0)
        f_teacher = f1_teacher.detach()
        # the four students as one batch of 4B against the B teacher samples (kernels index teacher / target with b % B)
        pl_all, d_all = ops.proto_loss(f1_stud, f_teacher, tcl)
        kl_all = crit.kl_cl(fuse_stud, pt_full, temp)
        for j in range(4):
            kl_all = kl_all + 0.5 ** (j + 1) * crit.kl_cl(preds_stud[j], pt_deep[j], temp, ups[j])
        pl_all, d_all, kl_all = pl_all.view(4, B, 1), d_all.view(4, B, 1), kl_all.view(4, B, 1)
        kl_loss, proto_loss, dist = (torch.zeros(B, 4, device=dev) for _ in range(3))
        for i in range(4):
            pl, d, kl = pl_all[i], d_all[i], kl_all[i]
            # single-present-modality sample: student == teacher exactly (see mmformer.py for the rationale)
            differs = (~(mask == onehots[i]).all(1)).float().view(B, 1)
            pl, d, kl = pl * differs, d * differs, kl * differs
            proto_loss = proto_loss + gates[i] * pl
            dist = dist + gates[i] * d
            kl_loss = kl_loss + gates[i] * kl
        return pred, prm_loss, sep_loss, kl_loss, proto_loss, dist
