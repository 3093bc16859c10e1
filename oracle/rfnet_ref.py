"""ORACLE (test infrastructure only -- never imported by the product path).

Plain-PyTorch restatement of rfnet.Model.forward (reference code/models/rfnet.py:15-403 with
general_conv3d / prm_generator*_pk / modal_fusion / region_fusion / region_aware_modal_fusion from
code/models/blocks.py:354-370, 396-416, 443-464, 495-531, 582-626).  Functional: driven by a dict with
the reference's state_dict keys.  Pinned against tests/golden/rfnet_*.npz (tools/make_golden.py).
"""
import torch
import torch.nn.functional as F

from .criterions_ref import (dice_loss_bs, prototype_passion_loss_bs, softmax_weighted_loss_bs,
                             temp_kl_loss_bs)

MODALS = ("flair", "t1ce", "t1", "t2")
NUM_CLS = 4


def _conv(P, name, x, stride=1):
    """nn.Conv3d; every 3x3x3 conv of RFNet is reflect padded (general_conv3d default, blocks.py:355)"""
    w, b = P[name + ".weight"], P[name + ".bias"]
    if w.shape[-1] == 3:
        x = F.pad(x, (1,) * 6, mode="reflect")
    return F.conv3d(x, w, b, stride=stride)


def _gconv(P, name, x, stride=1):
    """general_conv3d (blocks.py:366-370): Conv3d -> InstanceNorm -> LeakyReLU(0.2)"""
    return F.leaky_relu(F.instance_norm(_conv(P, name + ".conv", x, stride), eps=1e-5), 0.2)


def _up(x, k=2):
    return F.interpolate(x, scale_factor=k, mode="trilinear", align_corners=True) if k > 1 else x


def encoder(P, pre, x):
    """rfnet.py:35-48"""
    feats = []
    h = x
    for lvl in range(1, 5):
        h = _gconv(P, f"{pre}.e{lvl}_c1", h, stride=1 if lvl == 1 else 2)
        h = h + _gconv(P, f"{pre}.e{lvl}_c3", _gconv(P, f"{pre}.e{lvl}_c2", h))
        feats.append(h)
    return feats


def decoder_sep(P, xs):
    """rfnet.py:72-89"""
    pre = "decoder_sep"
    h = _gconv(P, f"{pre}.d3_c1", _up(xs[3]))
    for lvl in (3, 2, 1):
        h = torch.cat((h, xs[lvl - 1]), 1)
        h = _gconv(P, f"{pre}.d{lvl}_out", _gconv(P, f"{pre}.d{lvl}_c2", h))
        if lvl > 1:
            h = _gconv(P, f"{pre}.d{lvl - 1}_c1", _up(h))
    return F.softmax(_conv(P, f"{pre}.seg_layer", h), 1)


def _seq3(P, name, x):
    for i in range(3):
        x = _gconv(P, f"{name}.{i}", x)
    return x


def _prm(P, name, x1, x2_masked):
    """prm_generator_pk / _laststage_pk (blocks.py:409-416, 457-464): logits"""
    emb = _seq3(P, f"{name}.embedding_layer", x2_masked)
    h = emb if x1 is None else torch.cat((x1, emb), 1)
    return _conv(P, f"{name}.prm_layer.1", _gconv(P, f"{name}.prm_layer.0", h))


def _rfm(P, name, x, prm, mask):
    """region_aware_modal_fusion.forward (blocks.py:597-626); x (B,4,C,D,H,W), prm (B,4,D,H,W) detached"""
    B, K, C = x.shape[:3]
    y = x * mask[:, :, None, None, None, None].to(x.dtype)
    fused = []
    for i in range(NUM_CLS):
        pi = prm[:, i:i + 1, None]                                   # (B,1,1,D,H,W)
        region = y * pi                                              # modality features inside region i
        prm_avg = pi.mean((3, 4, 5)).expand(B, 1, C) + 1e-7          # blocks.py:507
        feat_avg = region.mean((3, 4, 5)) / prm_avg                  # (B,K,C)
        v = torch.cat((feat_avg.reshape(B, K * C), prm_avg[:, 0, 0:1]), 1)
        mf = f"{name}.modal_fusion.{i}.weight_layer"
        h = F.leaky_relu(F.linear(v, P[f"{mf}.0.weight"].flatten(1), P[f"{mf}.0.bias"]), 0.2)
        w = torch.sigmoid(F.linear(h, P[f"{mf}.2.weight"].flatten(1), P[f"{mf}.2.bias"]))   # (B,K)
        fused.append((region * w[:, :, None, None, None, None]).sum(1))
    rf = _seq3(P, f"{name}.region_fusion.fusion_layer", torch.cat(fused, 1))
    sc = _seq3(P, f"{name}.short_cut", y.flatten(1, 2))
    return torch.cat((rf, sc), 1)


def decoder_fuse(P, xs, mask):
    """rfnet.py:126-152.  xs: four (B,4,C,D,H,W) stacks"""
    pre = "decoder_fuse"
    m6 = mask[:, :, None, None, None, None]
    prm4 = _prm(P, f"{pre}.prm_generator4", None, (xs[3] * m6.to(xs[3].dtype)).flatten(1, 2))
    h = _rfm(P, f"{pre}.RFM4", xs[3], F.softmax(prm4, 1).detach(), mask)
    h = _gconv(P, f"{pre}.d3_c1", _up(h))
    prms, feats = {4: prm4}, {4: h}
    for lvl in (3, 2, 1):
        x = xs[lvl - 1]
        prm = _prm(P, f"{pre}.prm_generator{lvl}", h, (x * m6.to(x.dtype)).flatten(1, 2))
        prms[lvl] = prm
        r = _rfm(P, f"{pre}.RFM{lvl}", x, F.softmax(prm, 1).detach(), mask)
        h = _gconv(P, f"{pre}.d{lvl}_out", _gconv(P, f"{pre}.d{lvl}_c2", torch.cat((r, h), 1)))
        if lvl > 1:
            h = _gconv(P, f"{pre}.d{lvl - 1}_c1", _up(h))
        feats[lvl] = h
    logits = _conv(P, f"{pre}.seg_layer", feats[1])
    return logits, (prms[1], prms[2], prms[3], prms[4]), (feats[1], feats[2], feats[3], feats[4])


def _seg_losses(pred, target, up_op=None):
    return softmax_weighted_loss_bs(pred, target, NUM_CLS, up_op) + dice_loss_bs(pred, target, NUM_CLS, up_op=up_op)


def forward(P, x, mask, target=None, temp=1.0, mask_type="idt", use_passion=False, is_training=False):
    """rfnet.Model.forward (rfnet.py:217-403)"""
    B = x.shape[0]
    idt = mask_type != "pdt"
    m6 = mask[:, :, None, None, None, None]
    if idt:
        x = x * mask[:, :, None, None, None].to(x.dtype)
    enc = [encoder(P, f"{m}_encoder", x[:, i:i + 1]) for i, m in enumerate(MODALS)]
    xs = [torch.stack([enc[i][l] for i in range(4)], 1) for l in range(4)]
    if idt:
        xs = [s * m6.to(s.dtype) for s in xs]                                  # masker_nocat (rfnet.py:239-242)
    fuse_pred, preds, de_f_avg = decoder_fuse(P, xs, mask)
    pred = F.softmax(fuse_pred, 1)
    if not is_training:
        return pred
    sep_preds = [decoder_sep(P, [xs[l][:, i] if idt else enc[i][l] for l in range(4)]) for i in range(4)]
    if idt:
        sep_preds = [sp * mask[:, i, None, None, None, None].to(sp.dtype) for i, sp in enumerate(sep_preds)]
    onehot = torch.eye(4, dtype=torch.bool, device=x.device)
    gate = (lambda i: mask.to(x.dtype) * onehot[i][None].to(x.dtype)) if idt else \
           (lambda i: onehot[i][None].expand(B, 4).to(x.dtype))
    ups = [None] + [lambda t, k=k: _up(t, k) for k in (2, 4, 8)]                # up_ops = [Identity, x2, x4, x8]
    prm_loss = x.new_zeros(B, 1)
    for j, (pp, up) in enumerate(zip(preds, ups)):
        prm_loss = prm_loss + 0.5 ** (j + 1) * _seg_losses(F.softmax(pp, 1), target, up)
    sep_loss = x.new_zeros(B, 4)
    for i in range(4):
        sep_loss = sep_loss + gate(i) * _seg_losses(sep_preds[i], target)
    if not use_passion:
        return pred, prm_loss, sep_loss
    kl_loss, proto_loss, dist = x.new_zeros(B, 4), x.new_zeros(B, 4), x.new_zeros(B, 4)
    for i in range(4):
        m_i = onehot[i][None].expand(B, 4)
        fuse_i, preds_i, de_f_i = decoder_fuse(P, xs, m_i)                      # rfnet.py:269-275
        pl, d = prototype_passion_loss_bs(de_f_i[0], de_f_avg[0].detach(), target, fuse_i, fuse_pred.detach(), NUM_CLS, temp)
        proto_loss = proto_loss + gate(i) * pl
        dist = dist + gate(i) * d
        kl = temp_kl_loss_bs(fuse_i, fuse_pred.detach(), target, NUM_CLS, temp)
        for j, (pt_, ps_, up) in enumerate(zip(preds, preds_i, ups)):
            kl = kl + 0.5 ** (j + 1) * temp_kl_loss_bs(ps_, pt_.detach(), target, NUM_CLS, temp, up)
        kl_loss = kl_loss + gate(i) * kl
    return pred, prm_loss, sep_loss, kl_loss, proto_loss, dist
