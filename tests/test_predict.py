"""Sliding-window inference: CPU tests of the window / coverage / Dice logic against the oracle with a toy
model (incl. 2-rank window sharding over gloo), and a GPU test of the real mmformer product model against the model
oracle -- the "mean Dice vs ref" half of BASELINE.json's metric on synthetic data."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn.functional as F

from _util import build_product_model, small_weights
from oracle import mmformer_ref, predict_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Toy:
    """stand-in model: softmax of a fixed random 1x1x1 projection + a 3x3x3 blur (window-position dependent at borders)"""
    is_training = False

    def __init__(self):
        g = torch.Generator().manual_seed(4)
        self.w = torch.randn(4, 4, 3, 3, 3, generator=g) * 0.3

    def __call__(self, x, mask):
        return torch.softmax(F.conv3d(x * mask[:, :, None, None, None].float(), self.w, padding=1), 1)


def _vol(shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, 4, *shape, generator=g), torch.randint(0, 4, (1, *shape), generator=g)


@pytest.mark.parametrize("shape,patch", [((40, 36, 50), 16), ((16, 16, 16), 16), ((33, 20, 17), 16), ((48, 48, 31), 24)])
def test_windows_coverage_and_argmax_match_oracle(shape, patch):
    from passion_amd.utils import predict
    x, _ = _vol(shape)
    mask = torch.tensor([[True, True, False, True]])
    toy = Toy()
    p_ref, a_ref = predict_ref.test_volume_softmax(toy, x, mask, patch)
    for wpb in (1, 3):
        p, a = predict.sliding_window_predict(toy, x, mask, patch, windows_per_batch=wpb)
        assert torch.allclose(p, p_ref, atol=1e-6)
        assert torch.equal(a, a_ref)


def test_dice_matches_oracle_including_postprocessing():
    from passion_amd.utils import predict
    g = torch.Generator().manual_seed(1)
    for big in (False, True):
        out = torch.randint(0, 3, (2, 20, 20, 20), generator=g)
        tgt = torch.randint(0, 4, (2, 20, 20, 20), generator=g)
        out[0, :2 if not big else 12, :10, :10] = 3        # 200 (< 500: dropped) or 1200 enhancing voxels
        s, e = predict.softmax_output_dice_class4(out, tgt)
        s_r, e_r = predict_ref.softmax_output_dice_class4(out, tgt)
        assert torch.allclose(s, s_r) and torch.allclose(e, e_r)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from passion_amd import parallel
    from passion_amd.utils import predict
    ctx = parallel.init(backend="gloo")
    x, _ = _vol((40, 36, 50))
    p, a = predict.sliding_window_predict(Toy(), x, torch.ones(1, 4, dtype=torch.bool), 16, 2, ctx)
    if rank == 0:
        torch.save({"p": p, "a": a}, out)
    torch.distributed.destroy_process_group()


def test_window_sharding_over_two_ranks_equals_single(tmp_path):
    """windows dealt round-robin over 2 ranks + one all-reduce == serial loop (multi-GPU full-volume inference)"""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    x, _ = _vol((40, 36, 50))
    p_ref, a_ref = predict_ref.test_volume_softmax(Toy(), x, torch.ones(1, 4, dtype=torch.bool), 16)
    assert torch.allclose(got["p"], p_ref, atol=1e-6) and torch.equal(got["a"], a_ref)


@pytest.mark.gpu
def test_full_volume_mean_dice_vs_oracle_on_gpu():
    """real model: product (HIP) vs oracle (CPU) sliding-window masks on a 48x40x56 volume with 32^3 windows;
    mean Dice between the two masks must be 1 (bit-exact masks up to f32 ties)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from passion_amd.utils import predict
    dev = torch.device("cuda:0")
    S = 32
    m = build_product_model(S, small=True).to(dev)
    P = small_weights()
    shape = (48, 40, 56)
    x, tgt = _vol(shape, seed=5)
    mask = torch.tensor([[True, False, True, True]])

    def oracle_model(xw, mk):
        return mmformer_ref.forward(P, xw, mk, None, 1.0, "idt", False, False)

    p_ref, a_ref = predict_ref.test_volume_softmax(oracle_model, x, mask, S)
    p, a = predict.sliding_window_predict(m, x.to(dev), mask.to(dev), S, windows_per_batch=4)
    assert ((p.cpu() - p_ref).abs().max() / p_ref.abs().max()) < 1e-3
    agree = (a.cpu() == a_ref).float().mean().item()
    assert agree > 0.9999
    oh = lambda z: z  # masks are integer label maps already
    _, ev = predict.softmax_output_dice_class4(oh(a.cpu()), a_ref)
    assert float(ev[:, :3].mean()) > 0.9999        # mean Dice (WT, TC, ET) of product mask vs oracle mask
    _, ev_t = predict.softmax_output_dice_class4(a, tgt.to(dev))
    _, ev_r = predict_ref.softmax_output_dice_class4(a_ref, tgt)
    assert torch.allclose(ev_t.cpu(), ev_r, atol=2e-4)


def test_predict_runs_in_eval_mode_and_restores_the_mode():
    """reference predict.py:154/:251 wraps inference in model.eval() ... model.train(): dropout must be off inside"""
    from passion_amd.utils import predict

    class Drop(torch.nn.Module):
        is_training, num_cls = True, 4

        def __init__(self):
            super().__init__()
            self.toy, self.drop, self.modes = Toy(), torch.nn.Dropout(0.5), []

        def forward(self, x, mask):
            self.modes.append(self.training)
            return self.drop(self.toy(x, mask))

    x, _ = _vol((24, 20, 16))
    mask = torch.ones(1, 4, dtype=torch.bool)
    m = Drop().train()
    p, _ = predict.sliding_window_predict(m, x, mask, 16)
    p_ref, _ = predict_ref.test_volume_softmax(m.toy, x, mask, 16)
    assert m.modes and not any(m.modes) and m.training and m.is_training
    assert torch.allclose(p, p_ref, atol=1e-6)


# ---- pins from the reference itself (tools/make_golden_predict.py -> tests/golden/predict.npz) ---------------------------------
def test_dice_matches_the_references_own_function():
    """softmax_output_dice_class4 (predict.py:78-124) executed from the reference file in the build container: oracle and product
    restatements against its outputs, incl. the < 500-voxel post-processing branch (499 / 500 / 1200 voxels) and an absent class"""
    from _util import gold
    from passion_amd.utils import predict
    g = gold("predict.npz")
    for i in range(int(g["n_dice"])):
        out, tgt = torch.from_numpy(g[f"dice{i}_out"]).long(), torch.from_numpy(g[f"dice{i}_tgt"]).long()
        for fn in (predict_ref.softmax_output_dice_class4, predict.softmax_output_dice_class4):
            sep, ev = fn(out, tgt)
            assert np.allclose(np.asarray(sep), g[f"dice{i}_sep"], rtol=1e-6, atol=1e-7), (i, fn.__module__)
            assert np.allclose(np.asarray(ev), g[f"dice{i}_ev"], rtol=1e-6, atol=1e-7), (i, fn.__module__)


def _golden_volume(g):
    gen = torch.Generator().manual_seed(int(g["vol_seed"]))
    shape = tuple(int(v) for v in g["vol_shape"])
    x = torch.randn(1, 4, *shape, generator=gen)
    tgt = torch.randint(0, 4, (1, *shape), generator=gen)
    return x, tgt, torch.from_numpy(g["vol_mask"]), shape, int(g["vol_patch"])


def _check_volume(g, p, a, tol):
    shape = tuple(int(v) for v in g["vol_shape"])
    p, a = p.cpu(), a.cpu()
    assert float((p[:, :, shape[0] // 2] - torch.from_numpy(g["vol_pred_slice"])).abs().max()) < tol
    assert np.allclose(p.double().sum((2, 3, 4)).numpy(), g["vol_pred_sum"], rtol=tol)
    diff = np.nonzero(a.reshape(-1).numpy().astype(np.uint8) != g["vol_argmax"].reshape(-1))[0]
    assert np.isin(diff, g["vol_close_idx"]).all(), f"{len(diff)} voxels differ, not all of them reference-side ties < 1e-4"
    return diff


def test_oracle_window_loop_around_the_oracle_model_matches_the_reference_model():
    """oracle/predict_ref.py + oracle/mmformer_ref.py against the imported reference Model inside the same window loop"""
    from _util import gold
    g = gold("predict.npz")
    x, tgt, mask, shape, S = _golden_volume(g)
    P = small_weights()
    with torch.no_grad():
        p, a = predict_ref.test_volume_softmax(lambda xw, mk: mmformer_ref.forward(P, xw, mk, None, 1.0, "idt", False, False), x, mask, S)
    diff = _check_volume(g, p, a, 1e-5)
    assert len(diff) == 0
    _, ev = predict_ref.softmax_output_dice_class4(a, tgt)
    assert np.allclose(np.asarray(ev), g["vol_ev"], atol=1e-6)


@pytest.mark.gpu
def test_product_sliding_window_matches_the_reference_model_fixture():
    """the product's batched sliding-window predictor (HIP kernels, f32) against the fixture the imported reference Model
    produced: probabilities 1e-3, argmax bit-exact up to reference-side ties, Dice vs the label volume"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from _util import gold
    from passion_amd.utils import predict
    g = gold("predict.npz")
    dev = torch.device("cuda:0")
    x, tgt, mask, shape, S = _golden_volume(g)
    m = build_product_model(S, small=True).to(dev)
    p, a = predict.sliding_window_predict(m, x.to(dev), mask.to(dev), S, windows_per_batch=4)
    diff = _check_volume(g, p, a, 1e-3)
    print(f"product vs reference sliding-window mask: {len(diff)} of {a.numel()} voxels differ (all reference-side ties)")
    _, ev = predict.softmax_output_dice_class4(a, tgt.to(dev))
    assert np.allclose(ev.cpu().numpy(), g["vol_ev"], atol=2e-4)
