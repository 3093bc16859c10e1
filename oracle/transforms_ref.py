"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's training input pipeline, code/options.py:50:
    Compose([RandCrop3D(size), RandomRotion(10), RandomIntensityChange((0.1, 0.1)), RandomFlip(0), NumpyType((f32, i64))])
(code/data/transforms.py:86-120, 133-155, 217-250, 375-384, 408-431) followed by the dataset's layout change and one-hot
(code/data/datasets_nii.py:143-153).  Uses scipy.ndimage.rotate, like the reference.  Pinned against
tests/golden/augment.npz, which tools/make_golden.py produced by running the reference's own classes.
"""
import random

import numpy as np
from scipy.ndimage import rotate

ROT_AXES = [(1, 0), (2, 1), (2, 0)]          # transforms.py:90


def sample_params(shape, size, channels=4, angle_spectrum=10, factor=(0.1, 0.1)):
    """the random draws of one sample, in the order Compose makes them (transforms.py:414-431): every op's sample()
    first -- RandCrop3D (python `random`), RandomRotion, RandomFlip (numpy) -- then RandomIntensityChange inside tf(x)"""
    start = [random.randint(0, s - k) for k, s in zip(size, shape)]                     # :225
    axes = ROT_AXES[np.random.choice(list(range(len(ROT_AXES))))]                       # :95
    angle = np.random.randint(-angle_spectrum, angle_spectrum)                          # :96
    flips = [bool(np.random.choice([True, False])) for _ in range(3)]                   # :142-144
    shift = np.random.uniform(-factor[0], factor[0], size=[1, size[0], 1, 1, channels])     # :241
    scale = np.random.uniform(1.0 - factor[1], 1.0 + factor[1], size=[1, size[0], 1, 1, channels])
    return {"start": start, "size": list(size), "axes": axes, "angle": int(angle), "flips": flips,
            "shift": shift[0, :, 0, 0, :], "scale": scale[0, :, 0, 0, :]}


def apply(vol, seg, p, num_cls=4):
    """vol (H,W,D,C) float32, seg (H,W,D) integer -> x (C,h,w,d) float32, one-hot (num_cls,h,w,d) float64"""
    sl = tuple(slice(s, s + k) for s, k in zip(p["start"], p["size"]))
    x = vol[sl].copy()
    y = seg[sl].copy()
    x = np.stack([rotate(x[..., c], p["angle"], axes=p["axes"], reshape=False, order=0, mode="constant", cval=-1)
                  for c in range(x.shape[-1])], axis=-1)
    y = rotate(y, p["angle"], axes=p["axes"], reshape=False, order=0, mode="constant", cval=-1)
    x = x * p["scale"][:, None, None, :] + p["shift"][:, None, None, :]                 # float64, per (H index, channel)
    for ax, f in enumerate(p["flips"]):
        if f:
            x, y = np.flip(x, ax), np.flip(y, ax)
    x = x.astype(np.float32)
    y = y.astype(np.int64)
    xo = np.ascontiguousarray(x.transpose(3, 0, 1, 2))
    yo = np.eye(num_cls)[y.reshape(-1)].reshape(y.shape + (num_cls,))
    return xo, np.ascontiguousarray(yo.transpose(3, 0, 1, 2))
