"""ORACLE (test infrastructure only -- never imported by the product path).

Plain-PyTorch CPU restatement of the m2ftrans backbone, SURVEY.md 8(f) N4: reference code/models/m2ftrans.py:25-61 (MultiCrossToken),
:63-154 (Encoder / Decoder_sep), :157-254 (Decoder_fusion), :308-460 (masked transformer bottleneck, Weight_Attention), :473-771
(Model.forward with the four uni-modal PASSION students), with blocks from code/models/blocks.py:25-109 (DepthWiseConvBlock,
GroupConvBlock), :206-297 (MultiMaskCrossBlock / MultiMaskAttentionLayer), :354-370 (general_conv3d, post-norm), :481-493
(prm_fusion_pk), :567-580 (fusion_postnorm) and the attention masks of code/models/mask.py:5-36.
Functional: the weights arrive as a dict with the reference's state_dict keys.  PINNED by tests/golden/m2ftrans_*.npz, which
tools/make_golden_m2f.py wrote from the imported reference (tests/test_m2ftrans_oracle.py).

The token grid is a cube of edge p = S / 16 (the reference hard-codes patch_size = 5: S = 80)."""
import torch
import torch.nn.functional as F

from .criterions_ref import (dice_loss_bs, prototype_passion_loss_bs, softmax_weighted_loss_bs,
                             temp_kl_loss_bs)

MODALS = ("flair", "t1ce", "t1", "t2")
NUM_HEADS = 8
NUM_CLS = 4
DEPTH = 3


def _conv(P, name, x, stride=1, pad_mode="reflect"):
    w, b = P[name + ".weight"], P[name + ".bias"]
    if w.shape[-1] == 3:
        x = F.pad(x, (1,) * 6, mode="reflect" if pad_mode == "reflect" else "constant")
    return F.conv3d(x, w, b, stride=stride)


def _gconv(P, name, x, stride=1, pad_mode="reflect"):
    """general_conv3d (blocks.py:366-370): Conv3d -> InstanceNorm (eps 1e-5, no affine) -> LeakyReLU(0.2)"""
    return F.leaky_relu(F.instance_norm(_conv(P, name + ".conv", x, stride, pad_mode), eps=1e-5), 0.2)


def _up(x, k=2):
    return F.interpolate(x, scale_factor=k, mode="trilinear", align_corners=True)


def _ln_c(P, name, x, eps=1e-6):
    """LayerNorm over the channel axis of an NCDHW tensor, through the (B, L, C) view the reference takes (nchwd2nlc2nchwd,
    blocks.py:25-30).  The memory order matters for the pin: with the LayerNorm applied to a movedim() view instead, the forward
    agrees to 1e-6 but this random-init network amplifies that rounding into 1e-3 differences of its gradients."""
    t = F.layer_norm(x.flatten(2).transpose(1, 2), x.shape[1:2], P[name + ".weight"], P[name + ".bias"], eps)
    return t.transpose(1, 2).reshape(x.shape).contiguous()


def encoder(P, pre, x):
    """Encoder.forward (m2ftrans.py:87-104): post-norm convs, residual pair per level"""
    feats = []
    h = _gconv(P, f"{pre}.e1_c1", x)
    for lvl in range(1, 6):
        if lvl > 1:
            h = _gconv(P, f"{pre}.e{lvl}_c1", h, stride=2)
        h = h + _gconv(P, f"{pre}.e{lvl}_c3", _gconv(P, f"{pre}.e{lvl}_c2", h))
        feats.append(h)
    return feats


def decoder_sep(P, xs):
    """Decoder_sep.forward (m2ftrans.py:132-154) -> softmax probabilities"""
    pre = "decoder_sep"
    h = _gconv(P, f"{pre}.d4_c1", _up(xs[4]))
    for lvl in (4, 3, 2, 1):
        h = _gconv(P, f"{pre}.d{lvl}_out", _gconv(P, f"{pre}.d{lvl}_c2", torch.cat((h, xs[lvl - 1]), 1)))
        if lvl > 1:
            h = _gconv(P, f"{pre}.d{lvl - 1}_c1", _up(h))
    return F.softmax(_conv(P, f"{pre}.seg_layer", h), 1)


def depthwise_block(P, pre, x):
    """DepthWiseConvBlock (blocks.py:32-63): 1x1 -> LN -> GELU -> depthwise 3x3x3 (zero pad) -> LN -> GELU -> 1x1 -> LN"""
    x = F.gelu(_ln_c(P, f"{pre}.norm1", F.conv3d(x, P[f"{pre}.conv1.weight"], P[f"{pre}.conv1.bias"])))
    x = F.conv3d(x, P[f"{pre}.conv2.weight"], P[f"{pre}.conv2.bias"], padding=1, groups=x.shape[1])
    x = F.gelu(_ln_c(P, f"{pre}.norm2", x))
    return _ln_c(P, f"{pre}.norm3", F.conv3d(x, P[f"{pre}.conv3.weight"], P[f"{pre}.conv3.bias"]))


def group_block(P, pre, x, identity):
    """GroupConvBlock (blocks.py:65-109), proj_drop = 0: identity + (1x1 expand -> LN -> GELU -> grouped 3x3x3 -> LN -> GELU -> 1x1
    -> LN), then GELU"""
    e = x.shape[1]
    h = F.gelu(_ln_c(P, f"{pre}.norm1", F.conv3d(x, P[f"{pre}.pwconv1.weight"], P[f"{pre}.pwconv1.bias"])))
    h = F.conv3d(h, P[f"{pre}.dwconv.weight"], P[f"{pre}.dwconv.bias"], padding=1, groups=e)
    h = F.gelu(_ln_c(P, f"{pre}.norm2", h))
    h = _ln_c(P, f"{pre}.norm3", F.conv3d(h, P[f"{pre}.pwconv2.weight"], P[f"{pre}.pwconv2.bias"]))
    return F.gelu(identity + h)


def cross_attention(P, pre, query, feats, mask, drop_p=0.0):
    """MultiMaskAttentionLayer (blocks.py:246-297): channel attention of the query volume over the channels of the four modality
    volumes; the channels of an absent modality are masked out (mask_gen_cross4, mask.py:25-36)"""
    B, Cq = query.shape[:2]
    q = depthwise_block(P, f"{pre}.query_map", query).flatten(2)
    k = torch.cat([depthwise_block(P, f"{pre}.key_map_{m}", f).flatten(2) for m, f in zip(MODALS, feats)], 1)
    v = torch.cat([depthwise_block(P, f"{pre}.value_map_{m}", f).flatten(2) for m, f in zip(MODALS, feats)], 1)
    att = (q @ k.transpose(-2, -1)) * q.shape[-1] ** -0.5                       # (B, Cq, 4 C)
    keep = mask.repeat_interleave(k.shape[1] // 4, dim=1)[:, None, :]           # (B, 1, 4 C): modality-major channel blocks
    att = att.masked_fill(~keep, float("-inf")).softmax(-1)
    out = depthwise_block(P, f"{pre}.out_project", (att @ v).reshape(query.shape))
    return query + F.dropout(out, drop_p)


def cross_token(P, pre, feats, kernels, mask, drop_p=0.0):
    """MultiCrossToken (m2ftrans.py:25-61): two MultiMaskCrossBlocks (blocks.py:206-244); the first also runs the feature
    maps through ffn2 (ffn_feature_maps = i != num_layers - 1)"""
    feats = list(feats)
    for i in range(2):
        lp = f"{pre}.layers.{i}"
        kernels = cross_attention(P, f"{lp}.cross_attn", kernels, feats, mask, drop_p)
        kernels = group_block(P, f"{lp}.ffn1", kernels, kernels)
        if i == 0:
            feats = [group_block(P, f"{lp}.ffn2", f, f) for f in feats]
    return kernels


def fusion_postnorm(P, pre, stacked, mask):
    """fusion_postnorm (blocks.py:567-580): zero absent modalities, merge (K, C), 1x1 -> 3x3x3 -> 1x1 post-norm convs (reflect)"""
    y = (stacked * mask[:, :, None, None, None, None].to(stacked.dtype)).flatten(1, 2)
    for i in range(3):
        y = _gconv(P, f"{pre}.fusion_layer.{i}", y)
    return y


def prm_fusion(P, pre, x):
    """prm_fusion_pk (blocks.py:481-493) -> logits"""
    return _conv(P, f"{pre}.prm_layer.1", _gconv(P, f"{pre}.prm_layer.0", x))


def decoder_fusion(P, dx, fusion, mask, drop_p=0.0):
    """Decoder_fusion.forward (m2ftrans.py:218-254).  dx = (dx1, dx2, dx3 stacked (B,4,C,...), dx4, dx5 tuples of four volumes)"""
    pre = "decoder_fusion"
    dx1, dx2, dx3, dx4, dx5 = dx
    preds = {5: prm_fusion(P, f"{pre}.prm_fusion5", fusion)}
    h = torch.cat((cross_token(P, f"{pre}.CT5", dx5, fusion, mask, drop_p), fusion), 1)
    h = _gconv(P, f"{pre}.d5_out", _gconv(P, f"{pre}.d5_c2", h))
    h = _gconv(P, f"{pre}.d4_c1", _up(h))
    feats = {5: h}
    preds[4] = prm_fusion(P, f"{pre}.prm_fusion4", h)
    g = torch.cat((cross_token(P, f"{pre}.CT4", dx4, h, mask, drop_p), h), 1)
    g = _gconv(P, f"{pre}.d4_out", _gconv(P, f"{pre}.d4_c2", g))
    h = _gconv(P, f"{pre}.d3_c1", _up(g))
    feats[4] = h
    for lvl, dxl in ((3, dx3), (2, dx2), (1, dx1)):
        preds[lvl] = prm_fusion(P, f"{pre}.prm_fusion{lvl}", h)
        g = torch.cat((fusion_postnorm(P, f"{pre}.RFM{lvl}", dxl, mask), h), 1)
        g = _gconv(P, f"{pre}.d{lvl}_out", _gconv(P, f"{pre}.d{lvl}_c2", g))
        if lvl > 1:
            h = _gconv(P, f"{pre}.d{lvl - 1}_c1", _up(g))
            feats[lvl] = h
        else:
            feats[1] = g
    logits = _conv(P, f"{pre}.seg_layer", feats[1])
    # (de_x1, de_x2, de_x3, de_x4, de_x5) of the reference: the tensors its local names hold at the return (:254)
    return logits, tuple(preds[i] for i in (1, 2, 3, 4, 5)), (feats[1], feats[2], feats[3], feats[4], feats[5])


def fusion_mask(mask, n):
    """mask_gen_fusion (mask.py:5-22): (B, 5n, 5n) bool -- a modality's tokens attend to themselves only; the fusion tokens attend
    to the present modalities and to themselves"""
    B = mask.shape[0]
    allow = torch.zeros(B, 5, 5, dtype=torch.bool)
    for i in range(4):
        allow[:, i, i] = True
        allow[:, 4, i] = mask[:, i]
    allow[:, 4, 4] = True
    return allow.repeat_interleave(n, 1).repeat_interleave(n, 2)


def bottleneck(P, x5, mask, fusion, pos, drop_p=0.0):
    """Bottleneck.forward (m2ftrans.py:397-411) + MaskedTransformer depth 3 (:354-391, MaskedAttention :318-351).
    -> the five token sets (B, n, C) and the list of attention probabilities (B, heads, 5n, 5n)"""
    tok = torch.cat([f.flatten(2).transpose(1, 2) for f in x5] + [fusion], 1) + pos
    B, N, C = tok.shape
    allow = fusion_mask(mask, N // 5)[:, None]
    attns = []
    for j in range(DEPTH):
        a, f = f"Bottleneck.trans_bottle.cross_attention_list.{j}.fn", f"Bottleneck.trans_bottle.cross_ffn_list.{j}.fn"
        h = F.layer_norm(tok, (C,), P[f"{a}.norm.weight"], P[f"{a}.norm.bias"])
        qkv = F.linear(h, P[f"{a}.fn.qkv.weight"]).reshape(B, N, 3, NUM_HEADS, C // NUM_HEADS).permute(2, 0, 3, 1, 4)
        att = ((qkv[0] @ qkv[1].transpose(-2, -1)) * (C // NUM_HEADS) ** -0.5).masked_fill(~allow, float("-inf")).softmax(-1)
        att = F.dropout(att, drop_p)
        attns.append(att.detach())
        h = F.linear((att @ qkv[2]).transpose(1, 2).reshape(B, N, C), P[f"{a}.fn.proj.weight"], P[f"{a}.fn.proj.bias"])
        tok = tok + F.dropout(F.dropout(h, drop_p), drop_p)
        h = F.layer_norm(tok, (C,), P[f"{f}.norm.weight"], P[f"{f}.norm.bias"])
        h = F.dropout(F.gelu(F.linear(h, P[f"{f}.fn.net.0.weight"], P[f"{f}.fn.net.0.bias"])), drop_p)
        tok = tok + F.dropout(F.linear(h, P[f"{f}.fn.net.3.weight"], P[f"{f}.fn.net.3.bias"]), drop_p)
    return torch.chunk(tok, 5, 1), attns


def weight_attention(levels, tra, attn0, p):
    """Weight_Attention.forward (m2ftrans.py:420-460): the first layer's attention of the fusion tokens, summed over heads and
    over the fusion queries, is a (p, p, p) weight map per modality; it scales that modality's level-5 tokens and, nearest-
    upsampled x2 per level, its four finer feature maps"""
    B = tra[0].shape[0]
    n = p ** 3
    w = [attn0[:, :, 4 * n:, i * n:(i + 1) * n].sum(1).sum(-2).reshape(B, 1, p, p, p) for i in range(4)]
    out = {5: tuple(t * wi for t, wi in zip(tra, w))}
    for lvl in (4, 3, 2, 1):
        w = [F.interpolate(wi, scale_factor=2, mode="nearest") for wi in w]
        out[lvl] = tuple(f * wi for f, wi in zip(levels[lvl - 1], w))
    return out


def _seg_losses(pred, target, up_op=None):
    return softmax_weighted_loss_bs(pred, target, NUM_CLS, up_op) + dice_loss_bs(pred, target, NUM_CLS, up_op=up_op)


def forward(P, x, mask, target=None, temp=1.0, mask_type="idt", use_passion=False, is_training=False, drop_p=0.0):
    """m2ftrans.Model.forward (m2ftrans.py:512-771)"""
    B = x.shape[0]
    idt = mask_type != "pdt"
    mk = mask[:, :, None, None, None]
    if idt:
        x = x * mk.to(x.dtype)                                                 # :522-523
    enc = [encoder(P, f"{m}_encoder", x[:, i:i + 1]) for i, m in enumerate(MODALS)]
    if idt:                                                                      # :529-539: absent modalities' features are zero
        for i in range(4):
            enc[i] = [f * mask[:, i, None, None, None, None].to(f.dtype) for f in enc[i]]
    p = enc[0][4].shape[-1]
    levels = [[enc[i][l] for i in range(4)] for l in range(5)]
    x5 = levels[4]
    fusion = P["fusion"].expand(B, -1, -1)

    def fused_pass(m_pass):
        toks, attns = bottleneck(P, x5, m_pass, fusion, P["pos"], drop_p)
        vols = [t.reshape(B, p, p, p, -1).permute(0, 4, 1, 2, 3) for t in toks]
        w = weight_attention(levels, vols[:4], attns[0], p)
        dx = (torch.stack(w[1], 1), torch.stack(w[2], 1), torch.stack(w[3], 1), w[4], w[5])
        return decoder_fusion(P, dx, vols[4], m_pass, drop_p)

    fuse_pred, preds, de_f = fused_pass(mask)
    pred = F.softmax(fuse_pred, 1)
    if not is_training:
        return pred

    sep = [decoder_sep(P, enc[i]) for i in range(4)]
    if idt:
        sep = [s * mask[:, i, None, None, None, None].to(s.dtype) for i, s in enumerate(sep)]
    onehot = torch.eye(4, dtype=torch.bool)
    gate = (lambda i: onehot[i][None].expand(B, 4).to(x.dtype)) if not idt else \
           (lambda i: mask.to(x.dtype) * onehot[i][None].to(x.dtype))
    ups = [lambda t: t] + [lambda t, k=k: _up(t, k) for k in (2, 4, 8, 16)]     # up_ops = Identity, x2, x4, x8, x16 (:499-504)

    prm_loss = x.new_zeros(B, 1)
    for j, (pp, up) in enumerate(zip(preds, ups)):                              # :649-653: weights 1/2 .. 1/32
        prm_loss = prm_loss + 0.5 ** (j + 1) * _seg_losses(F.softmax(pp, 1), target, up if j else None)
    sep_loss = x.new_zeros(B, 4)
    for i in range(4):
        sep_loss = sep_loss + gate(i) * _seg_losses(sep[i], target)
    if not use_passion:
        return pred, prm_loss, sep_loss

    kl_loss, proto_loss, dist = x.new_zeros(B, 4), x.new_zeros(B, 4), x.new_zeros(B, 4)
    for i in range(4):                                                          # the four uni-modal students (:583-641)
        m_i = onehot[i][None].expand(B, 4)
        fuse_i, preds_i, de_f_i = fused_pass(m_i)
        pl, d = prototype_passion_loss_bs(de_f_i[0], de_f[0].detach(), target, fuse_i, fuse_pred.detach(), NUM_CLS, temp)
        proto_loss = proto_loss + gate(i) * pl
        dist = dist + gate(i) * d
        kl = temp_kl_loss_bs(fuse_i, fuse_pred.detach(), target, NUM_CLS, temp)
        for j, (pt_, ps_, up) in enumerate(zip(preds, preds_i, ups)):
            kl = kl + 0.5 ** (j + 1) * temp_kl_loss_bs(ps_, pt_.detach(), target, NUM_CLS, temp, up if j else None)
        kl_loss = kl_loss + gate(i) * kl
    return pred, prm_loss, sep_loss, kl_loss, proto_loss, dist
