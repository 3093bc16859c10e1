"""Input pipeline.  CPU: the oracle restatement reproduces what the reference's own transform classes
produced (tests/golden/augment.npz, tools/make_golden.py aug).  GPU: the product pipeline (passion_amd.data.transforms,
HIP kernel passion_augment) reproduces the same fixtures bit for bit, and the oracle on a 128^3-class crop."""
import random

import numpy as np
import pytest
import torch

from _util import gold


def _inputs(shape, seed):
    rs = np.random.RandomState(1000 + seed)
    vol = rs.randn(*shape, 4).astype(np.float32)
    seg = rs.randint(0, 4, shape).astype(np.uint8)
    return vol, seg


def test_oracle_matches_reference_transforms():
    from oracle import transforms_ref as T
    g = gold("augment.npz")
    for ci in range(int(g["ncases"])):
        shape, size, seed = tuple(g[f"c{ci}_shape"]), tuple(g[f"c{ci}_size"]), int(g[f"c{ci}_seed"])
        vol, seg = _inputs(shape, seed)
        random.seed(seed); np.random.seed(seed)
        p = T.sample_params(shape, size)
        x, yo = T.apply(vol, seg, p)
        assert x.dtype == np.float32 and yo.dtype == np.float64
        assert np.array_equal(x, g[f"c{ci}_x"]), f"case {ci}: x differs"
        assert np.array_equal(yo.argmax(0).astype(np.uint8), g[f"c{ci}_label"]), f"case {ci}: labels differ"
        assert np.array_equal(yo.sum((1, 2, 3)), g[f"c{ci}_onehot_sum"])


def test_scipy_rotation_affine_formula_all_angles():
    """the closed form the kernel uses == scipy.ndimage.rotate(order=0, cval=-1) for every angle / plane it can draw"""
    from scipy.ndimage import rotate
    from passion_amd.data.transforms import rotation_affine
    rs = np.random.RandomState(2)
    a = rs.randint(0, 50, (23, 40, 31)).astype(np.int64)
    for axes in [(1, 0), (2, 1), (2, 0)]:
        ax = sorted(axes)
        for angle in range(-10, 10):
            ref = rotate(a, angle, axes=axes, reshape=False, order=0, mode="constant", cval=-1)
            m00, m01, m10, m11, o0, o1 = rotation_affine(angle, a.shape[ax[0]], a.shape[ax[1]])
            i0, i1 = np.meshgrid(np.arange(a.shape[ax[0]], dtype=np.float64), np.arange(a.shape[ax[1]], dtype=np.float64), indexing="ij")
            c0 = o0 + i0 * m00; c0 = c0 + i1 * m01
            c1 = o1 + i0 * m10; c1 = c1 + i1 * m11
            ok = ~((c0 < 0) | (c0 > a.shape[ax[0]] - 1) | (c1 < 0) | (c1 > a.shape[ax[1]] - 1))
            j0 = np.floor(c0 + 0.5).astype(int).clip(0, a.shape[ax[0]] - 1)
            j1 = np.floor(c1 + 0.5).astype(int).clip(0, a.shape[ax[1]] - 1)
            am = np.moveaxis(a, ax, (0, 1))
            mine = np.moveaxis(np.where(ok[..., None], am[j0, j1], -1), (0, 1), ax)
            assert np.array_equal(mine, ref), (axes, angle)


@pytest.mark.gpu
def test_gpu_pipeline_matches_reference_fixtures():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from passion_amd.data import transforms as T
    g = gold("augment.npz")
    for ci in range(int(g["ncases"])):
        shape, size, seed = tuple(g[f"c{ci}_shape"]), tuple(int(v) for v in g[f"c{ci}_size"]), int(g[f"c{ci}_seed"])
        vol, seg = _inputs(shape, seed)
        random.seed(seed); np.random.seed(seed)
        tf = eval("T.Compose([T.RandCrop3D(size), T.RandomRotion(10), T.RandomIntensityChange((0.1,0.1)), T.RandomFlip(0), "
                  "T.NumpyType((np.float32, np.int64)),])")
        x, yo = tf([vol[None], seg[None]])
        assert x.is_cuda and x.dtype == torch.float32 and yo.dtype == torch.float64
        assert np.array_equal(x[0].cpu().numpy(), g[f"c{ci}_x"]), f"case {ci}: x differs"
        assert np.array_equal(yo[0].argmax(0).cpu().numpy().astype(np.uint8), g[f"c{ci}_label"]), f"case {ci}: labels differ"
        assert np.array_equal(yo[0].sum((1, 2, 3)).cpu().numpy(), g[f"c{ci}_onehot_sum"])
        from passion_amd import ops
        tcl = ops.target_to_cl(yo)          # the attached channels-last copy
        assert torch.equal(tcl.permute(0, 4, 1, 2, 3).double(), yo)


@pytest.mark.gpu
def test_gpu_pipeline_matches_oracle_large():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import transforms_ref as R
    from passion_amd.data import transforms as T
    shape, size = (100, 90, 80), (80, 64, 72)
    for seed in (1, 2, 3):
        vol, seg = _inputs(shape, seed)
        random.seed(seed); np.random.seed(seed)
        xr, yr = R.apply(vol, seg, R.sample_params(shape, size))
        random.seed(seed); np.random.seed(seed)
        tf = T.Compose([T.RandCrop3D(size), T.RandomRotion(10), T.RandomIntensityChange((0.1, 0.1)), T.RandomFlip(0),
                        T.NumpyType((np.float32, np.int64))])
        x, yo = tf([vol[None], seg[None]])
        assert np.array_equal(x[0].cpu().numpy(), xr)
        assert np.array_equal(yo[0].cpu().numpy(), yr)
        # signed labels: rotated-out voxels become -1 -> np.eye(4)[-1] = class 3
        random.seed(seed); np.random.seed(seed)
        xr2, yr2 = R.apply(vol, seg.astype(np.int16), R.sample_params(shape, size))
        random.seed(seed); np.random.seed(seed)
        x2, yo2 = tf([vol[None], seg.astype(np.int16)[None]])
        assert np.array_equal(yo2[0].cpu().numpy(), yr2) and not np.array_equal(yr2, yr)


@pytest.mark.gpu
def test_dataset_class_on_disk(tmp_path):
    """the reference's dataset contract (datasets_nii.py:95-160) on a tiny synthetic dataset: same RNG draws ->
    same sample as the oracle chain; returns GPU tensors, mask from the CSV"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import transforms_ref as R
    from passion_amd.data.datasets_nii import Brats_loadall_train_nii_idt, mask_array
    root = tmp_path
    (root / "vol").mkdir(); (root / "seg").mkdir()
    names, rows = [], []
    for i in range(3):
        vol, seg = _inputs((40, 44, 36), 50 + i)
        n = f"BraTS20_{i:03d}"
        np.save(root / "vol" / f"{n}_vol.npy", vol); np.save(root / "seg" / f"{n}_seg.npy", seg)
        names.append(n); rows.append(f"{n},{[3, 14, 7][i]},x,\"[3, 14]\"")
    (root / "split.csv").write_text("data_name,mask_id,mask,pos_mask_ids\n" + "\n".join(rows) + "\n")
    tfs = "Compose([RandCrop3D((32,32,32)), RandomRotion(10), RandomIntensityChange((0.1,0.1)), RandomFlip(0), NumpyType((np.float32, np.int64)),])"
    ds = Brats_loadall_train_nii_idt(transforms=tfs, root=str(root), num_cls=4, mask_type="idt", train_file=str(root / "split.csv"))
    assert len(ds) == 3
    random.seed(9); np.random.seed(9)
    x, yo, mask, name = ds[1]
    assert x.is_cuda and x.shape == (4, 32, 32, 32) and yo.shape == (4, 32, 32, 32) and yo.dtype == torch.float64
    assert name == names[1] and mask.cpu().tolist() == mask_array[14].tolist()
    vol, seg = _inputs((40, 44, 36), 51)
    random.seed(9); np.random.seed(9)
    xr, yr = R.apply(vol, seg, R.sample_params(vol.shape[:3], (32, 32, 32)))
    assert np.array_equal(x.cpu().numpy(), xr) and np.array_equal(yo.cpu().numpy(), yr)


@pytest.mark.gpu
def test_train_main_reads_npy_dataset(tmp_path):
    """python -m passion_amd.train without --synthetic: one short epoch from an on-disk dataset in the reference's format"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from passion_amd import train as ptrain
    root = tmp_path / "data"
    (root / "vol").mkdir(parents=True); (root / "seg").mkdir()
    rows = []
    for i in range(2):
        vol, seg = _inputs((40, 40, 40), 70 + i)
        n = f"BraTS20_{i:03d}"
        np.save(root / "vol" / f"{n}_vol.npy", vol); np.save(root / "seg" / f"{n}_seg.npy", seg)
        rows.append(f"{n},{[14, 6][i]},\"[True, True, True, True]\",\"[14]\"" if i == 0 else f"{n},6,\"[True, False, True, False]\",\"[6]\"")
    csv = tmp_path / "split.csv"
    csv.write_text("data_name,mask_id,mask,pos_mask_ids\n" + "\n".join(rows) + "\n")
    out = tmp_path / "out"
    ptrain.main(["--model", "mmformer", "--use_passion", "--num_epochs", "1", "--patch", "32", "--dtype", "bf16",
                 "--datapath", str(root), "--imbmrpath", str(csv), "--savepath", str(out)])
    ck = torch.load(out / "model_last.pth", map_location="cpu")
    assert ck["epoch"] == 0 and any(k.startswith("module.") for k in ck["state_dict"])
