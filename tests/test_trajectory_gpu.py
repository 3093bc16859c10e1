"""Multi-step parity against the imported reference (tools/make_golden_traj.py): K AdamW(amsgrad) steps on a volume whose
intensities carry the label, so the network learns and its prediction has MARGINS -- a random-init prediction is a near tie at
almost every voxel and cannot tell bf16 rounding from a small kernel bug.  The product replays the K steps from the
same seed (forward, train.py loss assembly, backward, FusedAdamW):

  * f32 mode: the loss of EVERY step within 1e-3 of the reference's trajectory, the final argmax mask bit-exact up to voxels the
    reference itself decides by < 1e-3, a few final weights within 5e-3;
  * bf16 mode = the benchmarked kernels (marching / lean bf16 convs, flash attention, side streams): every loss within 2 %,
    final-mask agreement and mean Dice (WT / TC / ET) against the reference's final mask above the stated floors."""
import numpy as np
import pytest
import torch

from _util import gold, mask_dice, rel, replay_trajectory, unpack2

pytestmark = pytest.mark.gpu

CASES = ["traj_plain_64.npz", "traj_passion_32.npz"]
# bf16 floors, measured on MI355X (see the printed line) with ~2x headroom on the disagreement
BF16 = {"traj_plain_64.npz": dict(loss=2e-2, agree=0.997, dice=0.997), "traj_passion_32.npz": dict(loss=2e-2, agree=0.995, dice=0.995)}


@pytest.mark.parametrize("name", CASES)
def test_f32_follows_the_reference_trajectory(name):
    g = gold(name)
    S = int(g["S"])
    m, losses, pred = replay_trajectory(g, torch.float32, torch.device("cuda:0"))
    ref = g["losses"]
    err = np.abs(losses - ref) / np.abs(ref)
    print(f"[{name} f32] per-step loss error: max {err.max():.2e} (step {int(err.argmax())}), final {err[-1]:.2e}")
    assert err.max() < 1e-3, err
    got = pred.argmax(1).reshape(-1).cpu().numpy().astype(np.uint8)
    want = unpack2(g["argmax_packed"], S ** 3)
    diff = np.nonzero(got != want)[0]
    print(f"[{name} f32] final mask: {len(diff)} of {S ** 3} voxels differ; reference has {len(g['close_idx'])} voxels with margin < 1e-3")
    # K optimizer steps amplify the f32 rounding differences between two implementations (the per-step losses above drift by up to
    # ~3e-4): a voxel may flip where the reference's own top-2 margin is small -- every differing voxel must lie in the
    # reference's list of margins < 2e-2, and they must be few
    assert np.isin(diff, g["close2_idx"]).all(), f"{(~np.isin(diff, g['close2_idx'])).sum()} differing voxels have a reference margin >= 2e-2"
    assert len(diff) <= 0.004 * S ** 3
    assert rel(pred[:, :, S // 2], g["pred_slice"]) < 3e-3
    sd = m.state_dict()
    for key in g.files:
        if key.startswith("final::"):
            assert rel(sd[key[7:]], g[key]) < 5e-3, (key, rel(sd[key[7:]], g[key]))


@pytest.mark.parametrize("name", CASES)
def test_bf16_benchmarked_path_keeps_the_trained_mask(name):
    g = gold(name)
    S = int(g["S"])
    m, losses, pred = replay_trajectory(g, torch.bfloat16, torch.device("cuda:0"))
    ref = g["losses"]
    err = np.abs(losses - ref) / np.abs(ref)
    got = pred.argmax(1).reshape(-1).cpu().numpy().astype(np.uint8)
    want = unpack2(g["argmax_packed"], S ** 3)
    dice, ndiff = mask_dice(got, want)
    agree = 1.0 - ndiff / S ** 3
    lab = unpack2(g["label_packed"], S ** 3)
    print(f"[{name} bf16] per-step loss error max {err.max():.2e}; final mask vs reference: {ndiff} of {S ** 3} voxels differ "
          f"(agreement {agree:.5f}), mean Dice WT/TC/ET {dice:.5f}; accuracy vs label {float((got == lab).mean()):.4f} "
          f"(reference {float(g['accs'][-1]):.4f})")
    t = BF16[name]
    assert err.max() < t["loss"], err
    assert agree > t["agree"] and dice > t["dice"], (agree, dice)
    diff = np.nonzero(got != want)[0]
    far = (~np.isin(diff, g["close3_idx"])).sum()
    assert far <= 0.0005 * S ** 3, f"{far} differing voxels where the reference's margin is >= 0.1"
    assert bool(torch.isfinite(pred).all())
