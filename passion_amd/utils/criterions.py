"""Per-sample losses of the PASSION hot path on the HIP kernels.

Public functions keep the reference's names, argument order and NCDHW tensors
(code/utils/criterions.py): softmax_weighted_loss_bs :59-76, dice_loss_bs :25-38,
temp_kl_loss_bs :92-103, prototype_passion_loss_bs :144-180 -- train.py:228 calls the first two on
the (B,4,D,H,W) prediction the Model returns.  The `_cl` variants take the model's internal
channels-last tensors.  `up_op` is an nn.Upsample(scale_factor=k, 'trilinear', align_corners=True)
as in the reference (or an int factor).
"""
import torch

from .. import ops

__all__ = ["softmax_weighted_loss_bs", "dice_loss_bs", "temp_kl_loss_bs", "prototype_passion_loss_bs"]


def _factor(up_op):
    if up_op is None:
        return 1
    if isinstance(up_op, int):
        return up_op
    sf = getattr(up_op, "scale_factor", None)
    if sf is None:
        return 1  # nn.Identity (rfnet.py:207)
    return int(sf[0] if isinstance(sf, (tuple, list)) else sf)


def seg_losses_cl(probs_cl, target_cl, up=1):
    """(B,2) = (weighted CE, Dice) of channels-last probabilities, optionally upsampled x`up` first"""
    if up > 1:           # the x k map is never materialised (ops.wce_dice_up: blended in registers, forward and backward)
        return ops.wce_dice_up(probs_cl, target_cl, up)
    return ops.wce_dice(probs_cl, target_cl, "cl")


def _seg_pair(output, target, up_op):
    tcl = ops.target_to_cl(target)
    k = _factor(up_op)
    if k == 1:
        box = getattr(output, "_passion_pair_box", None)      # Model.forward already ran this loss on a detached copy of its prediction
        if box is not None and box[0] is tcl and torch.is_grad_enabled():
            output._passion_pair_box = None
            return ops.DeferredFn.apply(output, box[1])      # (models/mmformer.py: beside the side streams' losses instead of behind them)
        return ops.wce_dice(output.float(), tcl, "ncdhw")
    return seg_losses_cl(_NcdhwToCl.apply(output.float()), tcl, k)


class _NcdhwToCl(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.to_channels_last(x)

    @staticmethod
    def backward(ctx, g):
        return ops.to_ncdhw(g.contiguous())


def seg_pair_bs(output, target, up_op=None):
    """softmax_weighted_loss_bs + dice_loss_bs in one pass -> (B,1) (the sum train.py:228 forms)"""
    return _seg_pair(output, target, up_op).sum(1, keepdim=True)


def softmax_weighted_loss_bs(output, target, num_cls=4, up_op=None):
    assert num_cls == 4 and output.shape[1] == 4
    return _seg_pair(output, target, up_op)[:, 0:1]


def dice_loss_bs(output, target, num_cls=4, eps=1e-7, up_op=None):
    assert num_cls == 4 and output.shape[1] == 4 and eps == 1e-7
    return _seg_pair(output, target, up_op)[:, 1:2]


def kl_cl(logit_s_cl, pt_cl, temp, up=1):
    """student logits (channels-last f32) vs the teacher's softmaxed probabilities, both at the student's resolution; up > 1: the
    loss is taken on the x`up` upsampled maps of both (mmformer.py:366-370 / criterions.py:92-103), inside the kernel"""
    ps = ops.softmax_classes(logit_s_cl, 1.0 / temp)
    if up > 1:
        return ops.kl_probs_up(ps, pt_cl, temp, up)
    return ops.kl_probs(ps, pt_cl, temp)


def temp_kl_loss_bs(logit_s, logit_t, target=None, num_cls=4, temp=1.0, up_op=None):
    k = _factor(up_op)
    ls, lt = _NcdhwToCl.apply(logit_s.float()), ops.to_channels_last(logit_t.detach().float())
    return kl_cl(ls, ops.softmax_classes(lt, 1.0 / temp), temp, k)


def prototype_passion_loss_bs(feature_s, feature_t, target, logit_s=None, logit_t=None, num_cls=4, temp=1.0,
                              up_op=None):
    fs = _NcdhwToCl.apply(feature_s)
    ft = ops.to_channels_last(feature_t.detach())
    return ops.proto_loss(fs, ft, ops.target_to_cl(target))
