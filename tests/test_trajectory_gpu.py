"""Multi-step parity against the imported reference (tools/make_golden_traj.py): K AdamW(amsgrad) steps on a volume whose
intensities carry the label, so the network learns and its prediction has MARGINS -- a random-init prediction is a near tie at
almost every voxel and cannot tell bf16 rounding from a small kernel bug.  The product replays the K steps from the
same seed (forward, train.py loss assembly, backward, FusedAdamW).

What is asserted, and why in this form:
  * the LOSS OF EVERY STEP against the reference's trajectory: f32 1e-3 (plain and PASSION; until round 4 the f32 kernels added their
    InstanceNorm-backward sums with float atomics and the PASSION bound had to be 4e-3 for a run-to-run spread of 7.8e-4 .. 1.7e-3:
    with 2^-40 fixed-point sums (conv_common.h epilogue_dgrad) three runs give the same 9.50e-4), bf16 (the benchmarked kernels:
    marching / lean bf16 convs, flash attention, side streams) 5e-3.  A wrong gradient anywhere bends the trajectory within a few
    steps; 30 steps within 2e-3 in bf16 is the strongest statement this suite makes about the benchmarked backward.
    PASSION's loss contains a DISCRETE gate (rp_mask, train.py:266-270: sign of a relative distance): the fixture stores the
    reference's gate of every step, and a step is compared only while every gate so far agreed (a flipped gate changes the
    loss by tens of percent and everything after it).  f32 runs free and never flips.  bf16 runs under the reference's gates
    (`rp_override`) so that all steps stay comparable; its OWN gates are recorded and may differ only where the reference's signed
    distance (fixture `rp_vals`) is a near tie (< 1e-2: e.g. step 4, modality t1ce: -0.0043).
  * the FINAL ARGMAX MASK: K steps amplify rounding differences of the trajectory itself (two f32 implementations drift ~1e-4
    in the weights), so voxels flip where the reference's own top-2 margin is small.  Asserted: (a) the differing voxels are few,
    (b) almost none of them is a voxel the reference decides by a margin >= 0.1 (fixture list close3_idx), (c) voxel accuracy
    against the LABEL equals the reference's to 2e-3 (the two masks are equally good), (d) mean Dice WT / TC / ET between the
    two masks above a floor.  After 20-30 steps the model is still unsure on ~1 % of the voxels (accuracy 0.88-0.94) and the
    enhancing-tumour class is ~1 % of the volume, which is what limits (d); the 128^3 fixture in bench.py's dice_vs_ref has larger
    structures."""
import numpy as np
import pytest
import torch

from _util import gold, mask_dice, rel, replay_trajectory, unpack2

pytestmark = pytest.mark.gpu

CASES = ["traj_plain_64.npz", "traj_passion_32.npz"]
# f32 PASSION: 20 steps amplify an f32 rounding difference of a gradient sum (4.6e-8 relative per op) to ~1e-3 of the loss: 9.5e-4 until
# round 6, 1.2e-3 since the three gradients of an encoder level are added in another order (conv_block_grouped pass_x = 2); every
# relative-preference gate still agrees with the reference's and the final mask is the same to 74 near-tie voxels of 32768
LOSS_TOL = {(torch.float32, False): 1e-3, (torch.float32, True): 2e-3, (torch.bfloat16, False): 5e-3, (torch.bfloat16, True): 1e-2}
# measured on MI355X (printed below): f32 311 / 262144 differing voxels (64^3, Dice 0.981), 81 / 32768 (32^3 PASSION, Dice 0.928: the
# enhancing class has ~60 voxels there); bf16 1848 / 262144 (17 with a reference margin >= 0.1, Dice 0.889)
# bf16 PASSION at 32^3 (all 21 steps under the reference's gates): 458 / 32768 differing voxels, 69 with a reference margin >= 0.1, Dice 0.791
# (the enhancing class has ~60 voxels), accuracy vs the label 0.8774 against the reference's 0.8776
MASK = {torch.float32: dict(diff=0.003, far=2e-4, acc=1e-3, dice=0.90), torch.bfloat16: dict(diff=0.02, far=2e-3, acc=4e-3, dice=0.80),
        (torch.bfloat16, True): dict(diff=0.02, far=4e-3, acc=4e-3, dice=0.75)}


def _gates(outs, mask):
    from passion_amd import train as ptrain
    return ptrain.relative_preference(outs[5], mask.float(), True).cpu().numpy()


GATE_TIE = 2e-2        # a gate the reference decides by a relative distance below this may flip under bf16 rounding (the gates of the
                       # replay are the reference's either way: this only bounds how close a decision the product's own gate may miss.
                       # Round 6: the first layer on the matrix cores sums its 27 taps in another order; the gates that differ moved
                       # from |distance| <= 0.0043 to one at 0.0155 in step 4+ of the 32^3 PASSION trajectory, per-step losses unchanged)


def _replay(name, dtype):
    g = gold(name)
    passion = bool(g["passion"])
    dev = torch.device("cuda:0")
    mask = torch.from_numpy(g["mask"]).to(dev)
    gates = []
    hook = (lambda it, outs, loss: gates.append(_gates(outs, mask))) if passion else None
    # bf16: the steps run under the REFERENCE's gates (the fixture holds them and the signed distances they are the sign of), the
    # product's own gates are recorded beside them: a gate may differ only where the reference's decision is a near tie
    force = g["rp_masks"] if (passion and dtype == torch.bfloat16) else None
    m, losses, pred = replay_trajectory(g, dtype, dev, on_step=hook, force_gates=force)
    return g, m, losses, pred, (np.stack(gates) if passion else None)


def _check(name, dtype):
    g, m, losses, pred, gates = _replay(name, dtype)
    S, passion = int(g["S"]), bool(g["passion"])
    tag = f"{name} {'f32' if dtype == torch.float32 else 'bf16'}"
    ref = g["losses"]
    err = np.abs(losses - ref) / np.abs(ref)
    upto = len(ref)
    if passion:
        differs = gates != g["rp_masks"]
        same = (~differs).all(1)
        first = int(np.argmin(same)) if not same.all() else len(ref)
        print(f"[{tag}] relative-preference gate agrees with the reference's for the first {first} of {len(ref)} steps; "
              f"{int(differs.sum())} of {differs.size} decisions differ, the reference's |distance| there: "
              f"{np.array2string(np.abs(g['rp_vals'][differs]), precision=4)}")
        if dtype == torch.float32:
            upto = first                                # free-running: comparable while every gate so far agreed
        else:                                           # gates forced to the reference's: every step is comparable
            assert (np.abs(g["rp_vals"][differs]) < GATE_TIE).all(), np.abs(g["rp_vals"][differs])
    print(f"[{tag}] per-step loss error over the comparable steps: max {err[:upto].max():.2e} (step {int(err[:upto].argmax())}); "
          f"all steps {np.array2string(err, precision=1, max_line_width=400)}")
    got = pred.argmax(1).reshape(-1).cpu().numpy().astype(np.uint8)
    want = unpack2(g["argmax_packed"], S ** 3)
    lab = unpack2(g["label_packed"], S ** 3)
    dice, ndiff = mask_dice(got, want)
    diff = np.nonzero(got != want)[0]
    far = int((~np.isin(diff, g["close3_idx"])).sum())
    acc, acc_ref = float((got == lab).mean()), float(g["accs"][-1])
    print(f"[{tag}] final mask vs reference: {ndiff} of {S ** 3} voxels differ ({far} of them where the reference's margin is >= 0.1), "
          f"mean Dice WT/TC/ET {dice:.5f}; accuracy vs label {acc:.4f} (reference {acc_ref:.4f})")
    t = MASK.get((dtype, passion), MASK[dtype])
    assert err[:upto].max() < LOSS_TOL[(dtype, passion)], err
    if dtype == torch.float32:
        assert upto == len(ref), "the f32 path flipped a relative-preference gate"
    if upto == len(ref):                               # same gates all the way: the final states are comparable
        assert ndiff <= t["diff"] * S ** 3 and far <= max(2, t["far"] * S ** 3), (ndiff, far)
        assert abs(acc - acc_ref) < t["acc"] and dice > t["dice"], (acc, acc_ref, dice)
    assert bool(torch.isfinite(pred).all())
    return g, m, pred


@pytest.mark.parametrize("name", CASES)
def test_f32_follows_the_reference_trajectory(name):
    g, m, pred = _check(name, torch.float32)
    S = int(g["S"])
    assert rel(pred[:, :, S // 2], g["pred_slice"]) < 5e-2      # probabilities after K steps of two f32 implementations: 2.4e-2 measured
    sd = m.state_dict()
    for key in g.files:
        # weights only: a conv bias in front of an InstanceNorm has an exactly zero gradient, what reaches AdamW is ~1e-9 of
        # rounding noise, and Adam turns noise of any size into steps of ~0.1 lr -- those biases random-walk in BOTH programs
        if key.startswith("final::") and not key.endswith("bias"):
            assert rel(sd[key[7:]], g[key]) < 1e-2, (key, rel(sd[key[7:]], g[key]))


@pytest.mark.parametrize("name", CASES)
def test_bf16_benchmarked_path_follows_the_reference_trajectory(name):
    _check(name, torch.bfloat16)
