"""The reference's training transform chain (code/options.py:50) with the per-voxel work on the GPU.

    Compose([RandCrop3D((S,S,S)), RandomRotion(10), RandomIntensityChange((0.1, 0.1)), RandomFlip(0),
             NumpyType((np.float32, np.int64)),])

Same class names and constructor arguments as code/data/transforms.py, so the string the reference eval()s
(datasets_nii.py:112) builds this pipeline unchanged.  The random draws are made on the host with the same
generators in the same order (python `random` for the crop, numpy for rotation / flips / intensity), so a seeded run
reproduces the reference's augmentation exactly; the cropped volume goes to the GPU once and ONE kernel
(passion_augment) does rotation (scipy.ndimage.rotate order-0 semantics, cval -1), intensity change, flips, the
HWDC -> CHWD layout change and the one-hot target -- the host side of the reference spends ~2 s per 128^3 sample in
scipy.ndimage.rotate alone.  Only this chain is supported: anything else raises NotImplementedError.
"""
import random
from ctypes import c_double, c_void_p

import numpy as np
import torch

from .. import _lib as L


class Base:
    def sample(self, *shape):
        return shape


class RandCrop3D(Base):
    def __init__(self, size):
        assert len(size) == 3
        self.size = list(size)

    def sample(self, *shape):
        self.start = [random.randint(0, s - k) for k, s in zip(self.size, shape)]      # transforms.py:225
        return self.size


class RandomRotion(Base):
    def __init__(self, angle_spectrum=10):
        assert isinstance(angle_spectrum, int)
        self.angle_spectrum = angle_spectrum
        self.axes = [(1, 0), (2, 1), (2, 0)]

    def sample(self, *shape):
        self.axes_buffer = self.axes[np.random.choice(list(range(len(self.axes))))]    # transforms.py:95-96
        self.angle_buffer = np.random.randint(-self.angle_spectrum, self.angle_spectrum)
        return list(shape)


class RandomIntensityChange(Base):
    def __init__(self, factor):
        self.shift, self.scale = factor
        assert self.shift > 0 and self.scale > 0

    def draw(self, n_h, channels):
        """the reference draws these inside tf(x), i.e. after every op's sample() (transforms.py:241-242)"""
        shift = np.random.uniform(-self.shift, self.shift, size=[1, n_h, 1, 1, channels])
        scale = np.random.uniform(1.0 - self.scale, 1.0 + self.scale, size=[1, n_h, 1, 1, channels])
        return shift[0, :, 0, 0, :], scale[0, :, 0, 0, :]


class RandomFlip(Base):
    def __init__(self, axis=0):
        self.axis = (1, 2, 3)

    def sample(self, *shape):
        self.x_buffer = bool(np.random.choice([True, False]))                          # transforms.py:142-144
        self.y_buffer = bool(np.random.choice([True, False]))
        self.z_buffer = bool(np.random.choice([True, False]))
        return list(shape)


class NumpyType(Base):
    def __init__(self, types, num=-1):
        self.types = types


def rotation_affine(angle_deg, n0, n1):
    """scipy.ndimage.rotate(reshape=False) -> the affine map handed to affine_transform: in = M out + offset"""
    a = np.deg2rad(angle_deg)
    c, s = np.cos(a), np.sin(a)
    rot = np.array([[c, s], [-s, c]], dtype=np.float64)
    n = np.array([n0, n1], dtype=np.float64)
    out_center = rot @ ((n - 1) / 2)
    off = (n - 1) / 2 - out_center
    return [rot[0, 0], rot[0, 1], rot[1, 0], rot[1, 1], off[0], off[1]]


class Compose(Base):
    """[RandCrop3D, RandomRotion, RandomIntensityChange, RandomFlip, NumpyType] only; `device` selects the GPU.
    __call__([x, y]) takes the reference's arrays -- x (1,H,W,D,C) float, y (1,H,W,D) integer (numpy) -- and returns
    GPU tensors already in the layout the dataset class produces (datasets_nii.py:143-153):
    x (1,C,h,w,d) float32 and the one-hot target (1,num_cls,h,w,d) float64."""

    def __init__(self, ops, device="cuda", num_cls=4):
        ops = list(ops)
        kinds = [type(o) for o in ops]
        if kinds != [RandCrop3D, RandomRotion, RandomIntensityChange, RandomFlip, NumpyType]:
            raise NotImplementedError("the GPU input pipeline implements the reference's train transform chain "
                                      "(options.py:50) only")
        self.crop, self.rot, self.inten, self.flip, _ = ops
        self.device = torch.device(device)
        self.num_cls = num_cls

    def __call__(self, img):
        x, y = img
        assert x.ndim == 5 and x.shape[0] == 1 and y.ndim == 4, "x (1,H,W,D,C), y (1,H,W,D)"
        shape = x.shape[1:4]
        for op in (self.crop, self.rot, self.inten, self.flip):       # Compose.sample (transforms.py:414-416)
            shape = op.sample(*shape)
        S0, S1, S2 = self.crop.size
        C = x.shape[4]
        shift, scale = self.inten.draw(S0, C)                         # drawn in tf(x), after all the samples
        sl = (0,) + tuple(slice(s, s + k) for s, k in zip(self.crop.start, self.crop.size))
        crop = torch.from_numpy(np.ascontiguousarray(x[sl], dtype=np.float32)).to(self.device, non_blocking=True)
        seg = torch.from_numpy(np.ascontiguousarray(y[sl]).astype(np.int32)).to(self.device, non_blocking=True)
        # scipy casts the rotation's cval = -1 to the label dtype: 0 for unsigned volumes (preprocess_brats.py:78 writes uint8)
        seg_cval = 0 if np.issubdtype(y.dtype, np.unsignedinteger) else -1
        return self.apply(crop, seg, sorted(self.rot.axes_buffer), int(self.rot.angle_buffer), shift, scale,
                          (self.flip.x_buffer, self.flip.y_buffer, self.flip.z_buffer), seg_cval)

    def apply(self, crop, seg, axes, angle, shift, scale, flips, seg_cval=0):
        """crop (S0,S1,S2,C) f32 / seg (S0,S1,S2) int32 on the GPU -> (x (1,C,...), onehot (1,ncls,...) f64); the
        channels-last f32 copy of the target that the loss kernels read is attached so that it is not rebuilt"""
        L.require_gpu(crop, seg)
        S0, S1, S2, C = crop.shape
        dev = crop.device
        sc = torch.from_numpy(np.ascontiguousarray(scale, dtype=np.float64)).to(dev, non_blocking=True)
        sh = torch.from_numpy(np.ascontiguousarray(shift, dtype=np.float64)).to(dev, non_blocking=True)
        xo = torch.empty(1, C, S0, S1, S2, dtype=torch.float32, device=dev)
        yo = torch.empty(1, self.num_cls, S0, S1, S2, dtype=torch.float64, device=dev)
        ycl = torch.empty(1, S0, S1, S2, self.num_cls, dtype=torch.float32, device=dev)
        n = (S0, S1, S2)
        rot6 = (c_double * 6)(*rotation_affine(angle, n[axes[0]], n[axes[1]]))
        L.check(L.load().passion_augment(c_void_p(crop.data_ptr()), c_void_p(seg.data_ptr()), S0, S1, S2, C, self.num_cls,
                                         axes[0], axes[1], rot6, int(seg_cval), int(flips[0]), int(flips[1]), int(flips[2]),
                                         c_void_p(sc.data_ptr()), c_void_p(sh.data_ptr()), c_void_p(xo.data_ptr()),
                                         c_void_p(yo.data_ptr()), c_void_p(ycl.data_ptr()), L.stream_ptr()), "passion_augment")
        try:
            yo._passion_cl = (yo._version, ycl)          # ops.target_to_cl finds it
        except Exception:
            pass
        return xo, yo
