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


def test_end_epoch_between_replays_is_seen_by_the_recording():
    """LossState.end_epoch() moves imb_beta once per epoch (train.py:325-335).  The recording reads imb_beta by ADDRESS: the update has
    to land in the tensor the recording knows (in place), and the replayed steps after it must equal the eager ones bit for bit."""
    from passion_amd import train as ptrain
    from passion_amd.graph import GraphedStep
    dev = torch.device("cuda:0")
    res = []
    for graphed in (False, True):
        m, opt, state = _make(True, False)
        x, target, _ = ptrain.synthetic_batch(1, S, dev, seed=3)
        mask = torch.tensor([[True, False, True, True]], device=dev)
        step = GraphedStep(m, opt, 4.0, state, warm=2) if graphed else (lambda a, b, c: ptrain.train_step(m, opt, a, b, c, 4.0, state))
        losses = []
        for it in range(8):
            losses.append(step(x, mask, target.view(target.shape)).clone())
            if it in (3, 5):
                ptr = state.imb_beta.data_ptr()
                state.end_epoch(epoch=it)
                assert state.imb_beta.data_ptr() == ptr                  # in place
        torch.cuda.synchronize()
        res.append((torch.stack(losses).double().cpu(), state.imb_beta.clone().cpu(), {k: v.detach().clone() for k, v in m.state_dict().items()}))
    (le, be, pe), (lg, bg, pg) = res
    assert not torch.equal(be, torch.ones(4))                            # the epoch update did move beta
    assert torch.equal(be, bg)
    assert torch.equal(le, lg), (le, lg)
    assert not [k for k in pe if not torch.equal(pe[k], pg[k])]


def test_rebound_loss_state_and_reloaded_optimizer_are_refused():
    from passion_amd import _lib as L
    from passion_amd import train as ptrain
    from passion_amd.graph import GraphedStep
    dev = torch.device("cuda:0")
    m, opt, state = _make(True, False)
    x, target, mask = ptrain.synthetic_batch(1, S, dev, seed=3)
    gs = GraphedStep(m, opt, 4.0, state, warm=1)
    for _ in range(3):
        gs(x, mask, target.view(target.shape))
    state.imb_beta = state.imb_beta.clone()                              # a NEW tensor: the recording would keep reading the old address
    with pytest.raises(L.PassionHipError):
        gs(x, mask, target.view(target.shape))
    m2, opt2, _ = _make(False, False)
    gs2 = GraphedStep(m2, opt2, 4.0, None, warm=0)                       # warm = 0 is raised to one eager step: AdamW's moments exist before the recording
    for _ in range(3):
        gs2(x, mask, target.view(target.shape))
    assert gs2.replays == 2
    opt2.load_state_dict(opt2.state_dict())                              # new moment tensors: the recording's pointers are stale
    with pytest.raises(L.PassionHipError):
        gs2(x, mask, target.view(target.shape))


def test_weights_changed_between_replays_are_repacked():
    """the recording re-packs the conv weight images behind ITS optimizer passes; a change from outside (load_state_dict) between two
    replays must reach the next replay's forward: the replayed run and the eager run see the same reload and stay bit-equal"""
    from passion_amd import train as ptrain
    from passion_amd.graph import GraphedStep
    dev = torch.device("cuda:0")
    res = []
    for graphed in (False, True):
        m, opt, _ = _make(False, False)
        x, target, mask = ptrain.synthetic_batch(1, S, dev, seed=3)
        sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
        step = GraphedStep(m, opt, 4.0, None, warm=2) if graphed else (lambda a, b, c: ptrain.train_step(m, opt, a, b, c, 4.0, None))
        losses = []
        for it in range(6):
            if it == 4:
                m.load_state_dict(sd0)                                   # back to the initial weights, behind the recording's back
            losses.append(step(x, mask, target.view(target.shape)).clone())
        torch.cuda.synchronize()
        res.append(torch.stack(losses).double().cpu())
    assert torch.equal(res[0], res[1]), res
    assert abs(float(res[1][4] - res[1][0])) < abs(float(res[1][3] - res[1][0]))     # step 4 really ran on the reloaded weights
