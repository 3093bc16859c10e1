"""Training datasets of the reference (code/data/datasets_nii.py:37-160) on top of the GPU input pipeline.

Same class names, constructor arguments, on-disk format (<root>/vol/<name>_vol.npy (H,W,D,4) float32,
<root>/seg/<name>_seg.npy (H,W,D) uint8; the imbalanced-missing-rate CSV with the columns data_name / mask_id / mask /
pos_mask_ids) and return value (x, one-hot target, modality mask, name) -- but `transforms` evaluates to
passion_amd.data.transforms.Compose, so x and the target come back as GPU tensors and the per-sample host work is the
npy read plus one crop copy.  Use with DataLoader(num_workers=0): the augmentation no longer needs worker processes
(and CUDA tensors cannot cross them); draw order of the random generators equals the reference's.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .transforms import (Compose, NumpyType, RandCrop3D, RandomFlip,  # noqa: F401  (names used by the eval()'d string)
                         RandomIntensityChange, RandomRotion)

# mask id -> present modalities [flair, t1ce, t1, t2]   (datasets_nii.py:27-30)
mask_array = np.array([[False, False, False, True], [False, True, False, False], [False, False, True, False],
                       [True, False, False, False], [False, True, False, True], [False, True, True, False],
                       [True, False, True, False], [False, False, True, True], [True, False, False, True],
                       [True, True, False, False], [True, True, True, False], [True, False, True, True],
                       [True, True, False, True], [False, True, True, True], [True, True, True, True]])

_MODAL = {"flair": [0], "t1ce": [1], "t1": [2], "t2": [3], "all": [0, 1, 2, 3]}


class _TrainBase(Dataset):
    def _setup(self, transforms, root, names, modal, num_cls, device):
        self.volpaths = [os.path.join(root, "vol", n + "_vol.npy") for n in names]
        self.names = names
        self.num_cls = num_cls
        self.modal_ind = _MODAL[modal]
        self.device = torch.device(device)
        self.transforms = eval(transforms) if isinstance(transforms, str) else transforms
        if not isinstance(self.transforms, Compose):
            raise NotImplementedError("pass the reference's train transform chain (options.py:50)")
        self.transforms.device, self.transforms.num_cls = self.device, num_cls

    def _sample(self, index):
        volpath = self.volpaths[index]
        x = np.load(volpath)
        y = np.load(volpath.replace("vol", "seg"))
        x, yo = self.transforms([x[None, ...], y[None, ...]])          # GPU: (1,C,h,w,d) f32, (1,ncls,h,w,d) f64
        cl = getattr(yo, "_passion_cl", None)
        x, yo = x[0][self.modal_ind], yo[0]
        if cl is not None:                                              # keep the channels-last copy findable after the squeeze
            try:
                yo._passion_cl_sample = cl[1][0]
            except Exception:
                pass
        return x, yo

    def __len__(self):
        return len(self.volpaths)


class Brats_loadall_train_nii_pdt(_TrainBase):
    """datasets_nii.py:37-93: mask id drawn uniformly from the 15 combinations per sample"""

    def __init__(self, transforms="", root=None, modal="all", num_cls=4, train_file="train.txt", device="cuda"):
        with open(os.path.join(root, train_file), "r") as f:
            names = sorted(i.strip() for i in f.readlines())
        self._setup(transforms, root, names, modal, num_cls, device)

    def __getitem__(self, index):
        x, yo = self._sample(index)
        mask_idx = np.random.choice(15, 1)
        mask = torch.squeeze(torch.from_numpy(mask_array[mask_idx]), dim=0).to(self.device)
        return x, yo, mask, self.names[index]


class Brats_loadall_train_nii_idt(_TrainBase):
    """datasets_nii.py:95-160: the per-subject mask of the imbalanced-missing-rate split (`idt`), a random choice among the
    subject's possible masks (`idt_drop`) or a uniform draw (`pdt`)"""

    def __init__(self, transforms="", root=None, modal="all", num_cls=4, mask_type="idt", train_file=None, device="cuda"):
        import pandas as pd
        excel = pd.read_csv(train_file)
        self.mask_ids = excel["mask_id"].values.tolist()
        self.samples_masks = excel["mask"].values.tolist()
        self.pos_mask_ids = excel["pos_mask_ids"].values.tolist()
        self.mask_type = mask_type
        self._setup(transforms, root, excel["data_name"].values.tolist(), modal, num_cls, device)

    def __getitem__(self, index):
        if self.mask_type == "idt":
            mask_idx = np.array([self.mask_ids[index]])
        elif self.mask_type == "idt_drop":
            mask_idx = np.random.choice(eval(self.pos_mask_ids[index]), 1)
        else:
            mask_idx = np.random.choice(15, 1)
        x, yo = self._sample(index)
        mask = torch.squeeze(torch.from_numpy(mask_array[mask_idx]), dim=0).to(self.device)
        return x, yo, mask, self.names[index]
