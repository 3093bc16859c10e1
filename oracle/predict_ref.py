"""ORACLE (test infrastructure only).  Serial restatement of the reference's sliding-window inference loop and Dice
(code/utils/predict.py:182-218, 78-124) around any `model(x_window, mask) -> softmax probabilities` callable.
predict.py itself cannot be imported here (medpy / nibabel / scipy.misc are absent).  PINNED (round 3) by
tests/golden/predict.npz (tools/make_golden_predict.py): the Dice function against the reference's OWN
softmax_output_dice_class4, taken out of the reference file's syntax tree and executed in the build container; the window loop
against the imported reference mmformer.Model driven through this restatement (the loop itself sits inside a function that
calls medpy's hd95 per subject and cannot run here: it stays a restatement of the cited lines)."""
import numpy as np
import torch


def test_volume_softmax(model, x, mask, patch_size):
    B, _, H, W, Z = x.shape
    lists = []
    for n in (H, W, Z):
        cnt = int(np.ceil((n - patch_size) / (patch_size * (1 - 0.5))))                 # :182
        idx = [i * int(patch_size * (1 - 0.5)) for i in range(0, cnt)]                   # :183-184
        idx.append(n - patch_size)                                                       # :185
        lists.append(idx)
    one = torch.ones(1, patch_size, patch_size, patch_size)
    weight1 = torch.zeros(1, 1, H, W, Z)
    for h in lists[0]:                                                                   # :198-203
        for w in lists[1]:
            for z in lists[2]:
                weight1[:, :, h:h + patch_size, w:w + patch_size, z:z + patch_size] += one
    pred = None
    for h in lists[0]:                                                                   # :209-214
        for w in lists[1]:
            for z in lists[2]:
                part = model(x[:, :, h:h + patch_size, w:w + patch_size, z:z + patch_size], mask)
                if pred is None:
                    pred = torch.zeros(B, part.shape[1], H, W, Z)
                pred[:, :, h:h + patch_size, w:w + patch_size, z:z + patch_size] += part
    pred = pred / weight1                                                                # :215
    return pred, torch.argmax(pred, dim=1)                                               # :218


def softmax_output_dice_class4(output, target):
    """predict.py:78-124"""
    eps = 1e-8
    res = {}
    o, t = {}, {}
    for c in (1, 2, 3):
        o[c], t[c] = (output == c).float(), (target == c).float()

    def dice(a, b):
        return (torch.sum(2 * (a * b), dim=(1, 2, 3)) + eps) / (torch.sum(a, dim=(1, 2, 3)) + torch.sum(b, dim=(1, 2, 3)) + eps)

    o4 = o[3] * 0.0 if torch.sum(o[3]) < 500 else o[3]
    sep = torch.stack((dice(o[1], t[1]), dice(o[2], t[2]), dice(o[3], t[3])), 1)
    ev = torch.stack((dice(o[1] + o[2] + o[3], t[1] + t[2] + t[3]), dice(o[1] + o[3], t[1] + t[3]), dice(o[3], t[3]),
                      dice(o4, t[3])), 1)
    return sep, ev
