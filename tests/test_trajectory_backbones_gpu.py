"""K-step parity for the other two backbones against the imported reference (tools/make_golden_traj_backbones.py): rfnet.Model and
m2ftrans.Model take 40 AdamW(amsgrad) PASSION steps at 32^3 from the committed start weights on the label-carrying volume of the
mmformer trajectories; the product replays them (forward, passion_amd.train loss assembly, backward, FusedAdamW).

Asserted: the loss of EVERY step against the reference's.  f32 runs free: a step is comparable while every relative-preference gate
(train.py:266-270, a discrete switch) so far agreed with the reference's, and the gates must agree wherever the reference decides by
more than GATE_TIE.  bf16 (the benchmarked kernels) runs under the reference's gates so that all 41 steps stay comparable; its own
gates may differ only at near ties.  The final mask: voxel accuracy against the label equal to the reference's, differing voxels few
outside the ones the reference itself decides by a margin < 0.1."""
import numpy as np
import pytest
import torch

from _util import gold, mask_dice, signal_volume, unpack2, build_m2ftrans

pytestmark = pytest.mark.gpu

# |relative distance| below which a gate may flip (tens of steps in: the trajectories themselves have drifted by ~1e-3 in f32 and by
# ~5e-3 in bf16, where WHICH near ties flip changes with every change of a summation order in the kernels: the largest flipped
# distance of m2ftrans was 0.026 .. 0.037 from build to build in round 6, 11 of 164 decisions)
GATE_TIE = {torch.float32: 3e-2, torch.bfloat16: 5e-2}
# measured on MI355X (printed by the tests): see the assertion messages
LOSS_TOL = {torch.float32: 2e-3, torch.bfloat16: 1e-2}


def _model(backbone, S, dev):
    if backbone == "rfnet":
        from passion_amd.models import rfnet
        from test_rfnet_oracle import rfnet_weights
        torch.manual_seed(1037)
        m = rfnet.Model(num_cls=4)
        m.load_state_dict(rfnet_weights(), strict=True)
    else:
        from test_m2ftrans_oracle import m2f_weights
        m = build_m2ftrans(S, small=True)
        m.load_state_dict(m2f_weights(), strict=True)
    m.eval()                                               # dropout off, as the fixture
    return m.to(dev)


def _replay(g, dtype, force):
    from passion_amd import train as ptrain
    dev = torch.device("cuda:0")
    backbone, S, K = str(g["backbone"]), int(g["S"]), int(g["K"])
    m = _model(backbone, S, dev)
    m.mask_type, m.use_passion, m.is_training, m.compute_dtype = "idt", True, True, dtype
    x, target = signal_volume(S, g["means"])
    x, target, mask = x.to(dev), target.to(dev), torch.from_numpy(g["mask"]).to(dev)
    opt = ptrain.make_optimizer(m, lr=float(g["lr"]), weight_decay=1e-4)
    losses, gates, pred = [], [], None
    for it in range(K + 1):
        outs = m(x, mask, target=target, temp=4.0)
        forced = torch.from_numpy(np.asarray(g["rp_masks"][it], dtype=np.float32)) if force else None
        loss = ptrain.assemble_loss(outs, target, mask, "idt", True, rp_override=forced)
        gates.append(ptrain.relative_preference(outs[5], mask.float(), True).cpu().numpy())
        losses.append(loss.detach())
        pred = outs[0].detach()
        if it == K:
            break
        opt.zero_grad()
        loss.backward()
        opt.step()
    return torch.stack(losses).double().cpu().numpy(), np.stack(gates), pred


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("name", ["traj_rfnet_passion_32.npz", "traj_m2ftrans_passion_32.npz"])
def test_backbone_follows_the_reference_trajectory(name, dtype):
    g = gold(name)
    S = int(g["S"])
    force = dtype == torch.bfloat16
    losses, gates, pred = _replay(g, dtype, force)
    ref = g["losses"]
    err = np.abs(losses - ref) / np.abs(ref)
    differs = gates != g["rp_masks"]
    same = (~differs).all(1)
    upto = len(ref) if (force or same.all()) else int(np.argmin(same))
    tag = f"{name} {'bf16' if force else 'f32'}"
    print(f"[{tag}] gates: {int(differs.sum())} of {differs.size} decisions differ (reference |distance| there: "
          f"{np.array2string(np.abs(g['rp_vals'][differs]), precision=4)}); comparable steps {upto} of {len(ref)}; "
          f"max loss error {err[:upto].max():.2e} at step {int(err[:upto].argmax())}")
    assert (np.abs(g["rp_vals"][differs]) < GATE_TIE[dtype]).all(), np.abs(g["rp_vals"][differs])       # a gate may flip only at a near tie
    assert upto >= min(len(ref), 10), upto
    assert err[:upto].max() < LOSS_TOL[dtype], err
    assert bool(torch.isfinite(pred).all())
    if upto == len(ref):
        got = pred.argmax(1).reshape(-1).cpu().numpy().astype(np.uint8)
        want, lab = unpack2(g["argmax_packed"], S ** 3), unpack2(g["label_packed"], S ** 3)
        dice, ndiff = mask_dice(got, want)
        diff = np.nonzero(got != want)[0]
        far = int((~np.isin(diff, g["close3_idx"])).sum())
        acc, acc_ref = float((got == lab).mean()), float(g["accs"][-1])
        print(f"[{tag}] final mask: {ndiff} of {S ** 3} voxels differ ({far} where the reference's margin is >= 0.1), mean Dice {dice:.4f}, "
              f"accuracy vs label {acc:.4f} (reference {acc_ref:.4f})")
        # (after 40 steps at 32^3 these models still decide most voxels by small margins -- accuracy 0.43 / 0.52 -- so the masks differ at
        # near ties: what is asserted is that hardly any voxel the reference decides by >= 0.1 differs, and that the two masks are equally good)
        # bf16: 40 steps of rounding noise on a model whose accuracy still moves by percents from step to step -- the mask is a sanity
        # bound there (measured: rfnet 0.4227 vs 0.4327, m2ftrans 0.4904 vs 0.5211), the per-step losses above are the statement
        assert abs(acc - acc_ref) < (5e-2 if force else 1e-2), (acc, acc_ref)
        # (bf16, m2ftrans: 1436 .. 3357 of 32768 from build to build -- round 6: 2545 -> 3357 when the marching input gradient cut its
        # depth segments finer on small launches, PASSION_MARCH_FILL, a change of summation order that also moved the accuracy
        # 0.4714 -> 0.4872 towards the reference's 0.5211)
        assert far <= (0.12 if force else 0.002) * S ** 3, (ndiff, far)
