"""CPU, world_size 2 over gloo: the N>1 path of passion_amd.parallel (parameter broadcast + bucketed
SUM all-reduce of gradients) equals single-process training on the global batch with summed
per-sample losses -- the reference's DataParallel semantics (train.py:90,229-231)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 3), nn.Tanh(), nn.Linear(3, 1))


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.randn(4, 6, generator=g), torch.randn(4, 1, generator=g)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from passion_amd import parallel
    ctx = parallel.init(backend="gloo")
    net = _net(100 + rank)                          # deliberately different init: broadcast must fix it
    red = parallel.GradReducer(net, ctx, bucket_mb=0)   # 0 -> one bucket per parameter: exercises many buckets
    x, y = _data()
    xs, ys = x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]    # weak scaling: each rank owns its samples
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    for _ in range(3):
        opt.zero_grad()
        ((net(xs) - ys) ** 2).sum().backward()     # per-sample losses are summed, never averaged
        red.all_reduce_grads()
        opt.step()
    t = parallel.all_reduce_max(ctx, float(rank))
    parallel.barrier(ctx)
    if rank == 0:
        torch.save({"sd": net.state_dict(), "max": t, "nb": len(red.buckets)}, out)
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_equals_global_batch(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["max"] == 1.0 and got["nb"] >= 6
    net = _net(100)
    x, y = _data()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    for _ in range(3):
        opt.zero_grad()
        ((net(x) - y) ** 2).sum().backward()
        opt.step()
    for k, v in net.state_dict().items():
        assert torch.allclose(v, got["sd"][k], atol=1e-6), k


def _rp_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from passion_amd import parallel
    from passion_amd import train as ptrain
    parallel.init(backend="gloo")
    d, mk = _rp_data()
    rp = ptrain.relative_preference(d[rank:rank + 1], mk[rank:rank + 1], idt=True)      # one sample per rank
    st = ptrain.LossState(torch.device("cpu"))
    st.epoch_dist_m += (d[rank:rank + 1] * mk[rank:rank + 1]).sum(0)
    rp_epoch = st.end_epoch(1)
    if rank == 0:
        torch.save({"rp": rp, "rp_epoch": rp_epoch, "beta": st.imb_beta}, out)
    torch.distributed.destroy_process_group()


def _rp_data():
    d = torch.tensor([[0.9, 0.1, 0.4, 0.2], [0.05, 0.6, 0.5, 0.3]])
    mk = torch.tensor([[1.0, 1.0, 0.0, 1.0], [1.0, 1.0, 1.0, 1.0]])
    return d, mk


def test_relative_preference_and_epoch_update_span_the_global_batch(tmp_path):
    """train.py:266-270 / :325-335 couple the samples of the whole (DataParallel-gathered) batch: two ranks with one sample
    each must get the rp_mask / imb_beta of one process holding both"""
    sys.path.insert(0, ROOT)
    from passion_amd import train as ptrain
    out = str(tmp_path / "rp.pt")
    mp.spawn(_rp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    d, mk = _rp_data()
    want = ptrain.relative_preference(d, mk, idt=True)
    assert torch.equal(got["rp"], want)
    solo = ptrain.relative_preference(d[:1], mk[:1], idt=True)
    assert not torch.equal(solo, want)                  # the fixture does distinguish local from global
    st = ptrain.LossState(torch.device("cpu"))
    st.epoch_dist_m += (d * mk).sum(0)
    assert torch.allclose(st.end_epoch(1), got["rp_epoch"]) and torch.allclose(st.imb_beta, got["beta"])


def _skip_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from passion_amd import parallel
    ctx = parallel.init(backend="gloo")
    torch.manual_seed(3)
    l1, l2, l3 = nn.Linear(6, 6), nn.Linear(6, 6), nn.Linear(6, 6)
    net = nn.ModuleList([l1, l2, l3])
    red = parallel.GradReducer(net, ctx, bucket_mb=0)
    x = torch.ones(2, 6) * (rank + 1)
    h = l1(x)
    if rank == 0:
        h = l2(h)                  # rank 1 never touches l2: its buckets complete only in all_reduce_grads()
    l3(h).sum().backward()
    red.all_reduce_grads()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in net.named_parameters()}, out)
    torch.distributed.destroy_process_group()


def test_buckets_go_out_in_the_same_order_on_every_rank(tmp_path):
    """a parameter without gradient on ONE rank must not reorder that rank's collectives (they would pair up with the wrong
    buckets of the other rank): launches are strictly by bucket index"""
    out = str(tmp_path / "skip.pt")
    mp.spawn(_skip_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(3)
    l1, l2, l3 = nn.Linear(6, 6), nn.Linear(6, 6), nn.Linear(6, 6)
    want = {}
    for rank in range(2):
        for m in (l1, l2, l3):
            m.zero_grad()
        x = torch.ones(2, 6) * (rank + 1)
        h = l1(x)
        if rank == 0:
            h = l2(h)
        l3(h).sum().backward()
        for i, m in enumerate((l1, l2, l3)):
            for n, p in m.named_parameters():
                if p.grad is not None:
                    want[f"{i}.{n}"] = want.get(f"{i}.{n}", 0) + p.grad.clone()
    for k, w in want.items():
        assert torch.allclose(got[k], w, atol=1e-6), k


def test_single_process_reducer_is_a_noop():
    sys.path.insert(0, ROOT)
    from passion_amd import parallel
    ctx = parallel.Ctx(0, 1, 0, torch.device("cpu"), "gloo")
    net = _net(1)
    red = parallel.GradReducer(net, ctx)
    net(torch.randn(2, 6)).sum().backward()
    g0 = [p.grad.clone() for p in net.parameters()]
    red.all_reduce_grads()
    assert all(torch.equal(a, p.grad) for a, p in zip(g0, net.parameters()))
