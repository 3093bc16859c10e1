"""Sliding-window full-volume inference + Dice on the GPU.

Counterpart of the reference's code/utils/predict.py: window placement :182-195 (edge `patch`, 50 % overlap,
one extra window flush with the far end of every axis), summation of the softmax outputs and division by
the coverage count :198-215, argmax :218, and softmax_output_dice_class4 :78-124 (WT / TC / ET /
ET-postprocessed, ET zeroed when < 500 voxels).  HD95 (medpy) is out of scope.

Differences: windows are batched through the model (samples are independent), the coverage count is the
outer product of three per-axis counts instead of a loop of `+= ones`, nothing leaves the device, and with
several ranks the windows are dealt round-robin and the accumulator is summed with one all-reduce (the
parity-faithful multi-GPU option for full volumes).
"""
import numpy as np
import torch


def window_starts(n, patch):
    """reference predict.py:182-185"""
    step = int(patch * (1 - 0.5))
    cnt = int(np.ceil((n - patch) / (patch * (1 - 0.5))))
    return [i * step for i in range(max(cnt, 0))] + [n - patch]


def coverage_count(shape, patch, device):
    """how many windows cover each voxel (reference predict.py:198-203) as an outer product"""
    axes = []
    for n in shape:
        c = torch.zeros(n, device=device)
        for s in window_starts(n, patch):
            c[s:s + patch] += 1
        axes.append(c)
    return axes[0][:, None, None] * axes[1][None, :, None] * axes[2][None, None, :]


@torch.no_grad()
def sliding_window_predict(model, x, mask, patch, windows_per_batch=4, ctx=None):
    """x (B,4,H,W,Z), mask (B,4) -> (prob (B,num_cls,H,W,Z) averaged over windows, argmax (B,H,W,Z) int64)"""
    B, _, H, W, Z = x.shape
    was, was_mode = model.is_training, getattr(model, "training", None)
    model.is_training = False
    if was_mode is not None:
        model.eval()                   # dropout off, as test_dice_hd95_softmax does (reference predict.py:154, restored :251)
    starts = [(h, w, z) for h in window_starts(H, patch) for w in window_starts(W, patch) for z in window_starts(Z, patch)]
    if ctx is not None and ctx.world > 1:
        starts = starts[ctx.rank::ctx.world]
    # allocated up front: a rank that was dealt no window (more ranks than windows) still joins the all-reduce below
    pred = torch.zeros(B, getattr(model, "num_cls", 4), H, W, Z, device=x.device, dtype=torch.float32)
    for i in range(0, len(starts), windows_per_batch):
        grp = starts[i:i + windows_per_batch]
        xin = torch.cat([x[:, :, h:h + patch, w:w + patch, z:z + patch] for h, w, z in grp], 0).contiguous()
        out = model(xin, mask.repeat(len(grp), 1))
        for j, (h, w, z) in enumerate(grp):
            pred[:, :, h:h + patch, w:w + patch, z:z + patch] += out[j * B:(j + 1) * B]
    if ctx is not None and ctx.world > 1:
        import torch.distributed as dist
        dist.all_reduce(pred, op=dist.ReduceOp.SUM)
    pred = pred / coverage_count((H, W, Z), patch, x.device)
    model.is_training = was
    if was_mode is not None:
        model.train(was_mode)
    return pred, torch.argmax(pred, dim=1)


def softmax_output_dice_class4(output, target):
    """reference predict.py:78-124 on integer masks (B,H,W,Z) -> (dice_separate (B,3), dice_evaluate (B,4)) tensors:
    separate = (ncr_net, edema, enhancing); evaluate = (whole, core, enhancing, enhancing_postpro)"""
    eps = 1e-8

    def dice(o, t):
        return ((2 * o * t).sum((1, 2, 3)) + eps) / (o.sum((1, 2, 3)) + t.sum((1, 2, 3)) + eps)

    o = [(output == c).float() for c in (1, 2, 3)]
    t = [(target == c).float() for c in (1, 2, 3)]
    o4 = o[2] * (o[2].sum() >= 500).float()        # post-processing: drop a tiny enhancing prediction (whole batch, as :104)
    sep = torch.stack((dice(o[0], t[0]), dice(o[1], t[1]), dice(o[2], t[2])), 1)
    ev = torch.stack((dice(o[0] + o[1] + o[2], t[0] + t[1] + t[2]), dice(o[0] + o[2], t[0] + t[2]), dice(o[2], t[2]),
                      dice(o4, t[2])), 1)
    return sep, ev
