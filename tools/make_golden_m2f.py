#!/usr/bin/env python
"""Golden fixtures for SURVEY 8(f) N4 (the m2ftrans backbone) from the IMPORTED reference (/root/reference/code, build container
only; same shims as tools/make_golden.py: Tensor.cuda -> identity for mask.py's / criterions.py's .cuda() literals, patch_size =
S // 16, eval() with is_training = True).  Small module constants (basic_dims 4 -> 64-wide bottleneck tokens, mlp_dim 128) and
float16-representable weights stored once (weights_m2f.npz), so parity does not depend on RNG order.

    python tools/make_golden_m2f.py      -> tests/golden/m2ftrans_*.npz, weights_m2f.npz
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402

GRAD_KEYS = ("flair_encoder.e1_c1.conv.weight", "t1_encoder.e3_c2.conv.weight", "Bottleneck.trans_bottle.cross_attention_list.1.fn.fn.qkv.weight",
             "decoder_fusion.CT5.layers.0.cross_attn.key_map_t1.conv2.weight", "decoder_fusion.CT4.layers.1.ffn1.dwconv.weight",
             "decoder_fusion.CT5.layers.0.ffn2.pwconv2.weight", "decoder_fusion.RFM2.fusion_layer.1.conv.weight",
             "decoder_fusion.d1_c2.conv.weight", "decoder_fusion.prm_fusion3.prm_layer.1.bias", "decoder_sep.d2_c1.conv.weight", "fusion", "pos")


def build(m2f, S):
    m2f.patch_size = S // 16
    m2f.basic_dims, m2f.mlp_dim = 4, 128
    torch.manual_seed(1037)
    m = m2f.Model(num_cls=4)
    g = torch.Generator().manual_seed(1038)
    with torch.no_grad():
        m.pos.copy_(torch.randn(m.pos.shape, generator=g) * 0.1)       # zero-initialised in the reference: give it values
        for p_ in m.parameters():
            p_.copy_(p_.half().float())
    m.eval()
    return m


def case(m2f, crit, tag, S, B, mask, mask_type, passion, blobs, temp=4.0):
    model = build(m2f, S)
    model.mask_type, model.use_passion, model.is_training = mask_type, passion, True
    x, target = mg.synth(B, S, blobs=blobs)
    mask_t = torch.tensor(mask, dtype=torch.bool).view(B, 4)
    outs = model(x, mask_t, target=target, temp=temp)
    pred = outs[0]
    fuse_bs = crit.softmax_weighted_loss_bs(pred, target, num_cls=4) + crit.dice_loss_bs(pred, target, num_cls=4)
    rec = {"S": S, "B": B, "mask": np.array(mask, dtype=bool).reshape(B, 4), "temp": temp, "mask_type": mask_type, "passion": passion,
           "blobs": blobs, "fuse_loss_bs": fuse_bs.detach().numpy(), "argmax": pred.argmax(1).numpy().astype(np.uint8),
           "pred_slice": pred[:, :, S // 2].detach().numpy(), "pred_sum": pred.detach().double().sum((2, 3, 4)).numpy()}
    for n, o in zip(["prm_loss", "sep_loss", "kl_loss", "proto_loss", "dist"], outs[1:]):
        rec[n] = o.detach().numpy()
    loss = fuse_bs.sum() + outs[1].sum()                                # train.py:228-280 with imb_beta = modal_weight = 1
    mk = mask_t.float() if mask_type != "pdt" else torch.ones(B, 4)
    if passion:
        sep_m, kl_m, proto_m = (outs[2] * mk).sum(0), (outs[3] * mk).sum(0), (outs[4] * mk).sum(0)
        rp = torch.zeros(4)
        for b in range(B):
            rp += mk[b] * (outs[5][b] / (outs[5][b].sum() / mk[b].sum()) - 1)
        rp_mask = (rp > 0).float()
        rec["rp_mask"] = rp_mask.numpy()
        loss = loss + (rp_mask * sep_m).sum() + 0.5 * kl_m.sum() + 0.1 * (rp_mask * proto_m).sum()
    else:
        loss = loss + (outs[2] * mk).sum()
    rec["loss"] = float(loss)
    model.zero_grad()
    loss.backward()
    gn = mg.grad_norms(model)
    rec["grad_keys"] = np.array(sorted(gn))
    rec["grad_norms"] = np.array([gn[k] for k in sorted(gn)])
    # the same step by the reference in FLOAT64 (model.double()): the arbiter for this network's f32 noise (two f32 runs of the
    # reference with different thread counts differ by up to 1.6e-4 in these norms; a GPU f32 run by several 1e-3 on the
    # random-label case)
    m64 = build(m2f, S).double()
    m64.mask_type, m64.use_passion, m64.is_training = mask_type, passion, True
    o64 = m64(x.double(), mask_t, target=target, temp=temp)
    f64 = crit.softmax_weighted_loss_bs(o64[0], target, num_cls=4) + crit.dice_loss_bs(o64[0], target, num_cls=4)
    l64 = f64.sum() + o64[1].sum()
    if passion:
        rp64 = torch.zeros(4, dtype=torch.float64)
        for b in range(B):
            rp64 += mk[b].double() * (o64[5][b] / (o64[5][b].sum() / mk[b].sum()) - 1)
        r64 = (rp64 > 0).double()
        l64 = l64 + (r64 * (o64[2] * mk).sum(0)).sum() + 0.5 * (o64[3] * mk).sum() + 0.1 * (r64 * (o64[4] * mk).sum(0)).sum()
    else:
        l64 = l64 + (o64[2] * mk).sum()
    l64.backward()
    g64 = mg.grad_norms(m64)
    rec["loss_f64"] = float(l64)
    rec["grad_norms_f64"] = np.array([g64[k] for k in sorted(gn)])
    sd = dict(model.named_parameters())
    for key in GRAD_KEYS:
        if sd[key].grad is not None:
            rec["grad::" + key] = sd[key].grad.numpy()
    wpath = os.path.join(mg.OUT, "weights_m2f.npz")
    if not os.path.exists(wpath):
        np.savez_compressed(wpath, **{k: v.numpy().astype(np.float16) for k, v in model.state_dict().items()})
    np.savez_compressed(os.path.join(mg.OUT, f"m2ftrans_{tag}.npz"), **rec)
    print("wrote m2ftrans", tag, "loss", rec["loss"], {k: np.round(rec[k], 4).tolist() for k in ("prm_loss", "sep_loss", "kl_loss", "dist") if k in rec})


def main():
    torch.set_num_threads(8)
    mg.import_reference()
    from models import m2ftrans
    from utils import criterions
    case(m2ftrans, criterions, "idt_passion_1011", 32, 1, [[1, 0, 1, 1]], "idt", True, True)
    case(m2ftrans, criterions, "idt_passion_b2", 32, 2, [[1, 1, 1, 1], [0, 0, 1, 0]], "idt", True, True)
    case(m2ftrans, criterions, "idt_plain_1111", 32, 1, [[1, 1, 1, 1]], "idt", False, False)
    case(m2ftrans, criterions, "pdt_passion_1101", 32, 1, [[1, 1, 0, 1]], "pdt", True, True)


if __name__ == "__main__":
    main()
