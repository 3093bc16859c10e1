"""passion_amd.graph.GraphedStep -- the launch mode bench.py measures by default -- against the eager train.train_step it records:
bit-equal losses and parameters over several iterations (dropout off), with the learning rate changed between replays and with
PASSION's device-side loss state (masks that change from step to step); with dropout on, reproducible from a seed; a call with another
signature is refused."""
import pytest
import torch

from _util import build_product_model

pytestmark = pytest.mark.gpu
S = 32


def _make(passion, dropout):
    from passion_amd import train as ptrain
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    m = build_product_model(S, small=False).to(dev)
    m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, torch.bfloat16
    m.train()
    if not dropout:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    opt = ptrain.make_optimizer(m, lr=1e-3)
    state = ptrain.LossState(dev) if passion else None
    return m, opt, state


def _run(passion, dropout, graphed, steps, lrs=None):
    from passion_amd import train as ptrain
    from passion_amd.graph import GraphedStep
    dev = torch.device("cuda:0")
    m, opt, state = _make(passion, dropout)
    x, target, _ = ptrain.synthetic_batch(1, S, dev, seed=3)
    masks = ([torch.tensor([[True, True, True, True]], device=dev), torch.tensor([[True, False, True, True]], device=dev),
              torch.tensor([[False, True, False, False]], device=dev)] if passion else [torch.ones(1, 4, dtype=torch.bool, device=dev)])
    step = GraphedStep(m, opt, 4.0, state, warm=2) if graphed else (lambda a, b, c: ptrain.train_step(m, opt, a, b, c, 4.0, state))
    losses = []
    for it in range(steps):
        if lrs is not None:
            for g in opt.param_groups:
                g["lr"] = lrs[it]
        losses.append(step(x, masks[it % len(masks)], target.view(target.shape)).clone())
    torch.cuda.synchronize()
    return torch.stack(losses).double().cpu(), {k: v.detach().clone() for k, v in m.state_dict().items()}, step


@pytest.mark.parametrize("passion", [False, True], ids=["plain", "passion"])
def test_replay_is_bit_equal_to_the_eager_step(passion):
    lrs = [1e-3, 1e-3, 1e-3, 5e-4, 5e-4, 2e-3, 2e-3, 1e-4]       # changed between replays (steps 0, 1 are the eager warm-up, 2 records)
    le, pe, _ = _run(passion, False, False, len(lrs), lrs)
    lg, pg, gs = _run(passion, False, True, len(lrs), lrs)
    assert gs.replays >= len(lrs) - 2
    assert torch.equal(le, lg), (le, lg)
    bad = [k for k in pe if not torch.equal(pe[k], pg[k])]
    assert not bad, bad[:5]


def test_dropout_under_replay_repeats_from_the_seed_and_differs_between_steps():
    l1, p1, _ = _run(False, True, True, 6)
    l2, p2, _ = _run(False, True, True, 6)
    assert torch.equal(l1, l2) and all(torch.equal(p1[k], p2[k]) for k in p1)
    le, _, _ = _run(False, False, True, 6)
    assert not torch.equal(l1[3:], le[3:])                       # the replays really draw dropout masks (and fresh ones: not the recording's)


def test_another_signature_is_refused():
    from passion_amd import _lib as L
    from passion_amd import train as ptrain
    dev = torch.device("cuda:0")
    _, _, gs = _run(False, False, True, 4)
    x, target, _ = ptrain.synthetic_batch(2, S, dev, seed=3)
    with pytest.raises(L.PassionHipError):
        gs(x, torch.ones(2, 4, dtype=torch.bool, device=dev), target)
