"""CPU tests of host-side logic that needs no kernel: the per-epoch subject order of train.main (every rank runs the same number
of steps), the brats_split mask distribution of BASELINE.json configs[2], bench.py's N > 1 self-launch, the 128^3 fixtures' own
consistency."""
import os
import sys

import numpy as np
import torch

from _util import flash_keep_mask, gold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_rank_gets_the_same_number_of_iterations_on_an_odd_dataset():
    from passion_amd import train as ptrain
    n, world, bs = 219, 8, 1                 # BraTS2020 split: 219 subjects on 8 GPUs (27 full global batches, 3 left over)
    iters = n // (bs * world)
    rows = [ptrain.epoch_order(n, 0, 1037, world, r, bs, iters) for r in range(world)]
    assert all(o.shape == (iters, bs) for o in rows)
    flat = np.concatenate([o.reshape(-1) for o in rows])
    assert len(set(flat.tolist())) == len(flat) == iters * bs * world      # disjoint subjects, nobody runs an extra step
    other = ptrain.epoch_order(n, 1, 1037, world, 0, bs, iters)
    assert not np.array_equal(rows[0], other)                                # reshuffled per epoch (DataLoader shuffle=True)
    tiny = ptrain.epoch_order(3, 0, 1037, 2, 1, 2, 1)
    assert tiny.shape == (1, 2)


def test_split_masks_follow_the_reference_csv():
    from passion_amd import train as ptrain
    g = gold("brats_split_mr2468.npz")
    assert g["mask_array"].tolist() == ptrain.MASK_TABLE                      # id -> mask table of datasets_nii.py:27-30
    assert np.bincount(g["mask_id"], minlength=15).tolist() == ptrain.SPLIT_MR2468_HIST
    assert g["modal_num"].tolist() == [90, 135, 184, 43]
    w = ptrain.split_modal_weight(1, 1)
    assert torch.allclose(w, torch.tensor([219 / 90, 219 / 135, 219 / 184, 219 / 43]))
    rng = np.random.RandomState(0)
    m = ptrain.sample_split_masks(rng, 20000).float().mean(0)
    assert torch.allclose(m, torch.tensor([90, 135, 184, 43.]) / 219, atol=0.02)
    g2 = gold("model_full128_idt_passion_split.npz")
    assert np.allclose(g2["modal_weight"], w.numpy(), rtol=1e-6) and abs(float((g2["imb_beta"] ** 2).sum()) - 4) < 1e-5
    assert g2["mask"].tolist()[0] in ptrain.MASK_TABLE and 2 <= g2["mask"].sum() < 4


def test_bench_self_launch_spawns_torchrun_before_touching_the_gpu(monkeypatch):
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
        raise AssertionError("self_launch must exit")
    except SystemExit as e:
        assert e.code == 7                                                    # the children's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not torch.cuda.is_initialized()


def test_fixture_argmax_packing_roundtrip():
    g = gold("model_full128_idt_plain_1111.npz")
    p = g["argmax_packed"]
    assert p.size * 4 == 128 ** 3
    lab = np.stack([(p >> (2 * j)) & 3 for j in range(4)], 1).reshape(-1)
    import hashlib
    assert hashlib.sha1(lab.astype(np.uint8).tobytes()).hexdigest() == str(g["argmax_sha1"])


def test_dropout_hash_statistics():
    """CPU-side property of the mask generator itself.  A word of the mask is mix(R(row) + C(key pair)) -- a row half and a column half
    that the kernels form once each (csrc/flash_attn.hip) -- so beside the keep rate the test holds what such a construction could
    get wrong: correlation along a row (every lag up to 40, and the power-of-two lags), between rows, along diagonals, between the two
    fields of one word, between seeds; at p = 0.1 (the reference's attn_drop) and at p = 0.5 (where a biased bit shows most)"""
    N = 1024
    for p, seed in ((0.1, 0x123456789abcdef), (0.5, 77)):
        keep, inv = flash_keep_mask(1, N, p, seed)
        k = keep.float().reshape(-1, N)                       # 8 heads x 1024 rows
        assert abs(float(k.mean()) - (1 - p)) < 2e-3
        assert abs(inv - 1.0 / (1.0 - round(p * 65536) / 65536.0)) < 1e-12
        kc = k - k.mean()
        v = float(kc.var())
        tol = 4e-3                                            # noise floor of one estimate: (8 * 1024 * 1024)^-1/2 = 3.5e-4
        for lag in list(range(1, 41)) + [64, 128, 256, 512]:
            assert abs(float((kc[:, :-lag] * kc[:, lag:]).mean()) / v) < tol, ("column lag", lag)
        for lag in list(range(1, 17)) + [32, 64, 1024, 2048]:
            assert abs(float((kc[:-lag] * kc[lag:]).mean()) / v) < tol, ("row lag", lag)
        for dr, dc in ((1, 1), (1, 2), (2, 1), (3, 5), (32, 32), (1, 16), (16, 1)):
            assert abs(float((kc[:-dr, :-dc] * kc[dr:, dc:]).mean()) / v) < tol, ("diagonal", dr, dc)
        assert abs(float((kc[:, 0::2] * kc[:, 1::2]).mean()) / v) < tol      # the two 16-bit fields of a word
        # no row and no column is favoured: the spread of their keep rates is the binomial one
        assert 0.9 < float(k.mean(1).std()) / (p * (1 - p) / N) ** 0.5 < 1.1
        assert 0.9 < float(k.mean(0).std()) / (p * (1 - p) / k.shape[0]) ** 0.5 < 1.1
        keep2, _ = flash_keep_mask(1, N, p, seed + 1)
        k2 = keep2.float().reshape(-1, N)
        assert abs(float(((k2 - k2.mean()) * kc).mean()) / v) < tol


def test_last_global_batch_is_wrap_padded_like_a_dataloader_without_drop_last():
    """reference train.py:122-128: DataLoader(shuffle=True) without drop_last -> ceil(N / batch) iterations per epoch"""
    from passion_amd import train as ptrain
    n, world, bs = 219, 8, 1
    iters = -(-n // (bs * world))            # 28
    rows = [ptrain.epoch_order(n, 0, 1037, world, r, bs, iters) for r in range(world)]
    assert all(o.shape == (iters, bs) for o in rows)
    flat = np.concatenate([o.reshape(-1) for o in rows])
    assert set(flat.tolist()) == set(range(n))                               # every subject is seen
    assert len(flat) == iters * world and len(flat) - n == 5                 # 5 wrap-around subjects fill the last global batch


def test_loss_state_round_trips_through_a_weights_only_checkpoint(tmp_path):
    """a resumed PASSION run must continue with the imb_beta / eta / modal_weight of the interrupted one"""
    from passion_amd import train as ptrain
    st = ptrain.LossState(torch.device("cpu"), modal_weight=torch.tensor([2.43, 1.62, 1.19, 5.09]))
    st.epoch_dist_m += torch.tensor([0.4, 0.1, 0.3, 0.2])
    st.end_epoch(100)                        # moves imb_beta, and eta (epoch % 100 == 0)
    assert abs(st.eta - 0.015) < 1e-12 and not torch.allclose(st.imb_beta, torch.ones(4))
    path = str(tmp_path / "ck.pth")
    torch.save({"epoch": 100, "loss_state": st.state_dict()}, path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    st2 = ptrain.LossState(torch.device("cpu"))
    st2.load_state_dict(ck["loss_state"])
    assert torch.equal(st2.imb_beta, st.imb_beta) and torch.equal(st2.modal_weight, st.modal_weight) and st2.eta == st.eta
    for s_ in (st, st2):
        s_.epoch_dist_m += torch.tensor([0.2, 0.3, 0.1, 0.4])
        s_.end_epoch(101)
    assert torch.equal(st2.imb_beta, st.imb_beta)
