"""ORACLE (test infrastructure only -- never imported by the product path).

Plain-PyTorch restatement of mmformer.Model.forward, the metric's hot path
(reference code/models/mmformer.py:24-659 with blocks from code/models/blocks.py:300-316, 533-542).
It is *functional*: weights come in as a dict that uses the reference's state_dict key names, so a
reference checkpoint / state_dict drives it directly.  Pinned against outputs of the imported
reference in tests/golden/ (tools/make_golden.py); used as the checker in tests/, in
__graft_entry__.smoke() and as the timed CPU baseline ("port") in bench.py.

Reference quirks that are reproduced on purpose:
  Q2  the T2 student masks its level-5 tokens with the T1 one-hot mask          (mmformer.py:522)
  Q3  (B,4N,512) inter-modal tokens are re-viewed as (B,p,p,p,2048)              (mmformer.py:444)
  Q4  fusion_prenorm's 3x3x3 conv is zero padded, all the others reflect         (blocks.py:301,538)
  Q9  idt masks the input before the encoders and the features again after       (mmformer.py:397-409)
"""
import torch
import torch.nn.functional as F

from .criterions_ref import (dice_loss_bs, prototype_passion_loss_bs, softmax_weighted_loss_bs,
                             temp_kl_loss_bs)

MODALS = ("flair", "t1ce", "t1", "t2")
NUM_HEADS = 8
NUM_CLS = 4


def _conv(P, name, x, stride=1, pad_mode="reflect"):
    """nn.Conv3d with the reference's padding_mode (kernel 1 or 3, padding k//2)."""
    w, b = P[name + ".weight"], P[name + ".bias"]
    if w.shape[-1] == 3:
        x = F.pad(x, (1,) * 6, mode="reflect" if pad_mode == "reflect" else "constant")
    return F.conv3d(x, w, b, stride=stride)


def _prenorm_conv(P, name, x, stride=1, pad_mode="reflect"):
    """general_conv3d_prenorm (blocks.py:312-316): InstanceNorm (biased var, eps 1e-5, no affine)
    -> LeakyReLU(0.2) -> Conv3d."""
    return _conv(P, name + ".conv", F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2), stride, pad_mode)


def _up(x, k=2):
    return F.interpolate(x, scale_factor=k, mode="trilinear", align_corners=True)


def encoder(P, pre, x):
    """Encoder.forward (mmformer.py:48-64)"""
    feats = []
    h = _conv(P, f"{pre}.e1_c1", x)
    for lvl in range(1, 6):
        if lvl > 1:
            h = _prenorm_conv(P, f"{pre}.e{lvl}_c1", h, stride=2)
        h = h + _prenorm_conv(P, f"{pre}.e{lvl}_c3", _prenorm_conv(P, f"{pre}.e{lvl}_c2", h))
        feats.append(h)
    return feats


def decoder_sep(P, xs):
    """Decoder_sep.forward (mmformer.py:93-114) -> softmax probabilities"""
    pre = "decoder_sep"
    h = _prenorm_conv(P, f"{pre}.d4_c1", _up(xs[4]))
    for lvl in (4, 3, 2, 1):
        h = torch.cat((h, xs[lvl - 1]), 1)
        h = _prenorm_conv(P, f"{pre}.d{lvl}_out", _prenorm_conv(P, f"{pre}.d{lvl}_c2", h))
        if lvl > 1:
            h = _prenorm_conv(P, f"{pre}.d{lvl - 1}_c1", _up(h))
    return F.softmax(_conv(P, f"{pre}.seg_layer", h), 1)


def _rfm(P, name, x):
    """fusion_prenorm (blocks.py:533-542): 1x1 -> 3x3x3 (zero pad) -> 1x1, each pre-normed"""
    x = _prenorm_conv(P, f"{name}.fusion_layer.0", x)
    x = _prenorm_conv(P, f"{name}.fusion_layer.1", x, pad_mode="zeros")
    return _prenorm_conv(P, f"{name}.fusion_layer.2", x)


def decoder_fuse(P, xs):
    """Decoder_fuse.forward (mmformer.py:155-189) -> logits, 4 deep-supervision logits, 5 feature maps"""
    pre = "decoder_fuse"
    f5 = _rfm(P, f"{pre}.RFM5", xs[4])
    preds = {4: _conv(P, f"{pre}.seg_d4", f5)}
    feats = {5: f5}
    h = _prenorm_conv(P, f"{pre}.d4_c1", _up(f5))
    for lvl in (4, 3, 2, 1):
        h = torch.cat((_rfm(P, f"{pre}.RFM{lvl}", xs[lvl - 1]), h), 1)
        f = _prenorm_conv(P, f"{pre}.d{lvl}_out", _prenorm_conv(P, f"{pre}.d{lvl}_c2", h))
        feats[lvl] = f
        if lvl > 1:
            preds[lvl - 1] = _conv(P, f"{pre}.seg_d{lvl - 1}", f)
            h = _prenorm_conv(P, f"{pre}.d{lvl - 1}_c1", _up(f))
    logits = _conv(P, f"{pre}.seg_layer", feats[1])
    return logits, (preds[1], preds[2], preds[3], preds[4]), tuple(feats[i] for i in (1, 2, 3, 4, 5))


def transformer(P, pre, x, pos, drop_p=0.0):
    """Transformer.forward depth 1 (mmformer.py:308-313): x+=pos; x+=drop(attn(LN x)); x+=FFN(LN x)"""
    a, f = f"{pre}.cross_attention_list.0.fn", f"{pre}.cross_ffn_list.0.fn"
    x = x + pos
    h = F.layer_norm(x, x.shape[-1:], P[f"{a}.norm.weight"], P[f"{a}.norm.bias"])
    B, N, C = h.shape
    qkv = F.linear(h, P[f"{a}.fn.qkv.weight"]).reshape(B, N, 3, NUM_HEADS, C // NUM_HEADS).permute(2, 0, 3, 1, 4)
    att = F.dropout(((qkv[0] @ qkv[1].transpose(-2, -1)) * (C // NUM_HEADS) ** -0.5).softmax(-1), drop_p)
    h = (att @ qkv[2]).transpose(1, 2).reshape(B, N, C)
    h = F.dropout(F.linear(h, P[f"{a}.fn.proj.weight"], P[f"{a}.fn.proj.bias"]), drop_p)
    x = x + F.dropout(h, drop_p)
    h = F.layer_norm(x, x.shape[-1:], P[f"{f}.norm.weight"], P[f"{f}.norm.bias"])
    h = F.dropout(F.gelu(F.linear(h, P[f"{f}.fn.net.0.weight"], P[f"{f}.fn.net.0.bias"])), drop_p)
    h = F.dropout(F.linear(h, P[f"{f}.fn.net.3.weight"], P[f"{f}.fn.net.3.bias"]), drop_p)
    return x + h


def mask_modal(x, mask):
    """MaskModal (mmformer.py:316-325): x (B,K,C,D,H,W), mask (B,K) bool -> zero absent modalities, merge K,C"""
    y = x * mask[:, :, None, None, None, None].to(x.dtype)
    return y.flatten(1, 2)


def _tokens(x):  # (B,C,p,p,p) -> (B,N,C)
    return x.flatten(2).transpose(1, 2)


def _untoken(t, p):  # (B,N,C) -> (B,C,pD,pH,pW); the reference's grid is a cube (mmformer.py:427), p = (pD, pH, pW) generalises it
    B, N, C = t.shape
    return t.transpose(1, 2).reshape(B, C, *p)


def _seg_losses(pred, target, up_op=None):
    return softmax_weighted_loss_bs(pred, target, NUM_CLS, up_op) + dice_loss_bs(pred, target, NUM_CLS, up_op=up_op)


def forward(P, x, mask, target=None, temp=1.0, mask_type="idt", use_passion=False, is_training=False,
            drop_p=0.0):
    """mmformer.Model.forward (mmformer.py:381-659).  Returns what the reference returns:
    softmax(fuse) | (pred, prm_loss, sep_loss) | (pred, prm_loss, sep_loss, kl_loss, proto_loss, dist)."""
    B = x.shape[0]
    if mask_type != "pdt":
        x = x * mask[:, :, None, None, None].to(x.dtype)                       # Q9 (mmformer.py:397-398)
    enc = [encoder(P, f"{m}_encoder", x[:, i:i + 1]) for i, m in enumerate(MODALS)]
    p = tuple(enc[0][4].shape[-3:])      # token grid; the reference hard-codes a cube of patch_size (mmformer.py:21,427,444)

    def stack(lvl):
        return torch.stack([enc[i][lvl] for i in range(4)], 1)                 # (B,4,C,D,H,W)

    xs = [mask_modal(stack(l), mask) for l in range(4)]
    if mask_type != "pdt":  # features of absent modalities are zeroed at all five levels (:405-415)
        for l in range(5):
            masked = stack(l) * mask[:, :, None, None, None, None].to(x.dtype)
            for i in range(4):
                enc[i][l] = masked[:, i]

    # intra-modal transformers (:417-430)
    pos = [P[f"{m}_pos"] for m in MODALS]
    intra = []
    for i, m in enumerate(MODALS):
        tok = _tokens(_conv(P, f"{m}_encode_conv", enc[i][4]))
        intra.append(_untoken(transformer(P, f"{m}_transformer", tok, pos[i], drop_p), p))
    intra = torch.stack(intra, 1)                                              # (B,4,512,p,p,p)
    intra = intra * mask[:, :, None, None, None, None].to(intra.dtype)         # x5_intra (:433)
    pos_cat = torch.cat(pos, 1)

    def inter(tok_mask=None):
        """(student: mask the intra features again,) concatenate tokens modality-major, inter-modal
        transformer, re-view (B,4N,512) as (B,p,p,p,2048) [Q3], 1x1x1 decode conv (:436-444, 474-482)"""
        feats = intra if tok_mask is None else intra * tok_mask[:, :, None, None, None, None].to(intra.dtype)
        tok = torch.cat([_tokens(feats[:, i]) for i in range(4)], 1)
        tok = transformer(P, "multimodal_transformer", tok, pos_cat, drop_p)
        vol = tok.reshape(B, *p, -1).permute(0, 4, 1, 2, 3)
        return _conv(P, "multimodal_decode_conv", vol)

    fuse_pred, preds, de_f_avg = decoder_fuse(P, xs + [inter()])
    pred = F.softmax(fuse_pred, 1)
    if not is_training:
        return pred

    sep_preds = [decoder_sep(P, enc[i]) for i in range(4)]
    if mask_type != "pdt":
        sep_preds = [sp * mask[:, i, None, None, None, None].to(sp.dtype) for i, sp in enumerate(sep_preds)]
    onehot = torch.eye(4, dtype=torch.bool, device=x.device)
    gate = (lambda i: onehot[i][None].expand(B, 4).to(x.dtype)) if mask_type == "pdt" else \
           (lambda i: mask.to(x.dtype) * onehot[i][None].to(x.dtype))
    ups = [lambda t, k=k: _up(t, k) for k in (2, 4, 8, 16)]

    prm_loss = x.new_zeros(B, 1)
    for j, (pp, up) in enumerate(zip(preds, ups)):                             # (:540-544)
        prm_loss = prm_loss + 0.5 ** (j + 1) * _seg_losses(F.softmax(pp, 1), target, up)
    sep_loss = x.new_zeros(B, 4)
    for i in range(4):
        sep_loss = sep_loss + gate(i) * _seg_losses(sep_preds[i], target)
    if not use_passion:
        return pred, prm_loss, sep_loss

    kl_loss, proto_loss, dist = x.new_zeros(B, 4), x.new_zeros(B, 4), x.new_zeros(B, 4)
    for i in range(4):                                                          # students (:469-531, 592-633)
        m_i = onehot[i][None].expand(B, 4)
        xs_i = [mask_modal(torch.stack([enc[j][l] for j in range(4)], 1), m_i) for l in range(4)]
        tok_mask = onehot[2][None].expand(B, 4) if i == 3 else m_i              # Q2 (mmformer.py:522)
        fuse_i, preds_i, de_f_i = decoder_fuse(P, xs_i + [inter(tok_mask)])
        pl, d = prototype_passion_loss_bs(de_f_i[0], de_f_avg[0].detach(), target, fuse_i, fuse_pred.detach(),
                                          NUM_CLS, temp)
        proto_loss = proto_loss + gate(i) * pl
        dist = dist + gate(i) * d
        kl = temp_kl_loss_bs(fuse_i, fuse_pred.detach(), target, NUM_CLS, temp)
        for j, (pt_, ps_, up) in enumerate(zip(preds, preds_i, ups)):
            kl = kl + 0.5 ** (j + 1) * temp_kl_loss_bs(ps_, pt_.detach(), target, NUM_CLS, temp, up)
        kl_loss = kl_loss + gate(i) * kl
    return pred, prm_loss, sep_loss, kl_loss, proto_loss, dist
