"""Depth sharding (BASELINE.json configs[4]), CPU / gloo, world 2: the exchange logic of passion_amd.depth on plain tensors --
halo planes (reflect / zero at the global ends, one or two low planes), their adjoint, and the D gather -- against the same
operation on the whole tensor."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


D_FULL = 8


def _full(seed=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(2, D_FULL, 5, 6, 4, generator=g, dtype=torch.float64)


def _pad_d(x, mode):
    """one plane of reflect / zero padding on both ends of dim 1"""
    if mode == "reflect":
        return torch.cat([x[:, 1:2], x, x[:, -2:-1]], 1)
    return torch.cat([torch.zeros_like(x[:, :1]), x, torch.zeros_like(x[:, :1])], 1)


def _worker(rank, world, port, out, d_full=8):
    global D_FULL
    D_FULL = d_full
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from passion_amd import depth
    shard = depth.DepthCtx()
    x = _full()
    Dl = x.shape[1] // world
    res = {}
    for name, reflect, n_lo, n_hi in (("reflect11", True, 1, 1), ("zero11", False, 1, 1), ("reflect20", True, 2, 0)):
        xl = x[:, rank * Dl:(rank + 1) * Dl].clone().requires_grad_(True)
        ext = depth.HaloExtendFn.apply(xl, shard, reflect, n_lo, n_hi)
        w = torch.arange(ext.numel(), dtype=torch.float64).view(ext.shape) * (rank + 1) * 1e-3
        (ext * w).sum().backward()
        res[name] = (ext.detach().clone(), xl.grad.clone(), w)
    xl = x[:, rank * Dl:(rank + 1) * Dl].clone().requires_grad_(True)
    full = depth.GatherDFn.apply(xl, shard)
    (full * (rank + 2.0)).pow(2).sum().backward()
    res["gather"] = (full.detach().clone(), xl.grad.clone())
    torch.save(res, f"{out}.{rank}")
    torch.distributed.destroy_process_group()


import pytest


@pytest.mark.parametrize("d_full", [8, 2])        # 2: ONE plane per rank (level 5 of a 128^3 volume on 8 GPUs): the mirrored plane is the neighbour's
def test_halo_exchange_and_gather_match_the_whole_tensor(tmp_path, d_full):
    global D_FULL
    D_FULL = d_full
    out = str(tmp_path / "d")
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), out, d_full), nprocs=world, join=True)
    got = [torch.load(f"{out}.{r}") for r in range(world)]
    x = _full()
    Dl = x.shape[1] // world
    for name, mode, n_lo, n_hi in (("reflect11", "reflect", 1, 1), ("zero11", "constant", 1, 1), ("reflect20", "reflect", 2, 0)):
        xr = x.clone().requires_grad_(True)
        xp = _pad_d(xr, mode)                                                                  # (B, D + 2, H, W, C)
        total = 0
        for r in range(world):
            lo = r * Dl + 1 - (n_lo if n_lo == 1 else 1)          # padded index of the first halo plane of rank r
            ref_ext = xp[:, lo:lo + 1 + Dl + n_hi]
            if n_lo == 2:
                ref_ext = torch.cat([torch.zeros_like(ref_ext[:, :1]), ref_ext], 1)
            ext, _, w = got[r][name]
            assert torch.allclose(ext, ref_ext.detach()), (name, r)
            total = total + (ref_ext * w).sum()
        total.backward()
        for r in range(world):
            assert torch.allclose(got[r][name][1], xr.grad[:, r * Dl:(r + 1) * Dl]), (name, r)
    xr = x.clone().requires_grad_(True)
    sum((xr * (r + 2.0)).pow(2).sum() for r in range(world)).backward()
    for r in range(world):
        assert torch.equal(got[r]["gather"][0], x)
        assert torch.allclose(got[r]["gather"][1], xr.grad[:, r * Dl:(r + 1) * Dl])


def test_single_rank_context_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from passion_amd import depth
    shard = depth.DepthCtx(rank=0, world=1)
    x = _full().requires_grad_(True)
    ext = depth.HaloExtendFn.apply(x, shard, True, 1, 1)
    ref = _pad_d(x, "reflect")
    assert torch.equal(ext, ref)
    g = torch.randn(ext.shape, dtype=torch.float64)
    (gx,) = torch.autograd.grad(ext, x, g)
    (gr,) = torch.autograd.grad(ref, x, g)
    assert torch.allclose(gx, gr)
