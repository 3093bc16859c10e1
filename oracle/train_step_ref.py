"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference training loop's per-iteration loss assembly
(code/train.py:222-280 PASSION branch, :410-440 plain branch) around oracle.mmformer_ref.forward,
and of the epoch-level imbalance update (train.py:325-335).  Pinned by the `loss` / `rp_mask` /
gradient entries of tests/golden/model_*.npz, which tools/make_golden.py computed with the same
assembly around the imported reference Model.
"""
import torch

from . import mmformer_ref
from .criterions_ref import dice_loss_bs, softmax_weighted_loss_bs


def assemble_loss(outs, target, mask, mask_type="idt", use_passion=True, imb_beta=None, modal_weight=None,
                  warmup=False):
    """-> (loss, parts dict).  train.py:228-280 (passion) / :413-440 (plain)."""
    pred = outs[0]
    B = pred.shape[0]
    one = pred.new_ones(4)
    imb_beta = one if imb_beta is None else imb_beta
    modal_weight = one if modal_weight is None else modal_weight
    fuse_loss = (softmax_weighted_loss_bs(pred, target, 4) + dice_loss_bs(pred, target, 4)).sum()
    prm_loss = outs[1].sum()
    mk = mask.to(pred.dtype)
    idt = mask_type != "pdt"
    red = (lambda t: (t * mk).sum(0)) if idt else (lambda t: t.sum(0))
    sep_m = red(outs[2])
    parts = {"fuse": fuse_loss, "prm": prm_loss, "sep_m": sep_m}
    if not use_passion:
        sep = sep_m.sum()
        loss = sep if warmup else fuse_loss + sep + prm_loss
        parts["loss"] = loss
        return loss, parts
    kl_m, proto_m, dist_bs = red(outs[3]), red(outs[4]), outs[5]
    rp = pred.new_zeros(4)
    for b in range(B):  # relative preference, train.py:239-242 / 265-268
        if idt:
            rp = rp + mk[b] * (dist_bs[b] / (dist_bs[b].sum() / mk[b].sum()) - 1)
        else:
            rp = rp + (dist_bs[b] / dist_bs[b].mean() - 1)
    rp_mask = (rp > 0).to(pred.dtype)
    w = modal_weight if idt else one
    kl = (imb_beta * w * kl_m).sum()
    proto = (rp_mask * w * proto_m).sum()
    if warmup:
        loss = (imb_beta * w * sep_m).sum()
    else:
        loss = fuse_loss + (rp_mask * imb_beta * w * sep_m).sum() + prm_loss + 0.5 * kl + 0.1 * proto
    parts.update(kl_m=kl_m, proto_m=proto_m, dist_m=red(dist_bs), rp_mask=rp_mask, loss=loss)
    return loss, parts


def imb_beta_update(imb_beta, epoch_dist_m, eta):
    """train.py:325-335: beta <- clamp(beta - eta * rp_epoch, 0.1, 4), renormalised to |beta|_2 = 2"""
    avg = epoch_dist_m.sum() / 4.0
    rp_epoch = (avg - epoch_dist_m) / avg
    beta = torch.clamp(imb_beta - eta * rp_epoch, 0.1, 4.0)
    return 2 * beta / beta.pow(2).sum().sqrt(), rp_epoch


def train_step(P, x, mask, target, temp=4.0, mask_type="idt", use_passion=True, drop_p=0.0):
    """forward + loss assembly; P values must require grad for backward"""
    outs = mmformer_ref.forward(P, x, mask, target, temp, mask_type, use_passion, True, drop_p)
    loss, parts = assemble_loss(outs, target, mask, mask_type, use_passion)
    return loss, outs, parts
