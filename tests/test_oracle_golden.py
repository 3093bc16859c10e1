"""CPU: pin the oracle (oracle/) against the golden vectors produced by the imported reference
(tools/make_golden.py), and check that the product Model's construction reproduces the reference's
parameter set (names, shapes, RNG stream)."""
import numpy as np
import pytest
import torch

from _util import build_product_model, gold, grad_norms, rel, small_weights, synth
from oracle import criterions_ref as C
from oracle import mmformer_ref, train_step_ref


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_losses_match_reference():
    g = gold("losses.npz")
    logits, logits_t = _t(g["logits"]), _t(g["logits_t"])
    probs = torch.softmax(logits, 1)
    for tn, key in (("rand", "target"), ("blob", "target_b")):
        t = _t(g[key]).double()
        assert rel(C.softmax_weighted_loss_bs(probs, t, 4), g[f"wce_{tn}"]) < 1e-6
        assert rel(C.dice_loss_bs(probs, t, 4), g[f"dice_{tn}"]) < 1e-6
        pl, d = C.prototype_passion_loss_bs(_t(g["fs"]), _t(g["ft"]), t, None, None, 4, 4.0)
        assert rel(pl, g[f"proto_{tn}"]) < 1e-5 and rel(d, g[f"protodist_{tn}"]) < 1e-5
    t = _t(g["target"]).double()
    assert rel(C.temp_kl_loss_bs(logits, logits_t, t, 4, 4.0), g["kl_T4"]) < 1e-6
    assert rel(C.temp_kl_loss_bs(logits, logits_t, t, 4, 1.0), g["kl_T1"]) < 1e-6
    for k in (2, 4, 8):
        up = torch.nn.Upsample(scale_factor=k, mode="trilinear", align_corners=True)
        lo, lo_t = _t(g[f"lo{k}"]), _t(g[f"lo{k}_t"])
        assert rel(C.softmax_weighted_loss_bs(torch.softmax(lo, 1), t, 4, up), g[f"wce_up{k}"]) < 1e-6
        assert rel(C.dice_loss_bs(torch.softmax(lo, 1), t, 4, up_op=up), g[f"dice_up{k}"]) < 1e-6
        assert rel(C.temp_kl_loss_bs(lo, lo_t, t, 4, 4.0, up), g[f"kl_up{k}"]) < 1e-6


def _run_oracle(g, P):
    S, B = int(g["S"]), int(g["B"])
    x, target = synth(B, S, blobs=bool(g["blobs"]))
    mask = _t(g["mask"])
    passion, mask_type = bool(g["passion"]), str(g["mask_type"])
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss, outs, parts = train_step_ref.train_step(P, x, mask, target, float(g["temp"]), mask_type, passion)
    return P, loss, outs, parts


def _check_outputs(g, outs, tol=2e-5):
    pred = outs[0]
    S = int(g["S"])
    assert rel(pred[:, :, S // 2], g["pred_slice"]) < tol
    assert rel(pred.double().sum((2, 3, 4)), g["pred_sum"]) < tol
    assert (pred.argmax(1).numpy().astype(np.uint8) == g["argmax"]).mean() > 0.9999
    for n, o in zip(["prm_loss", "sep_loss", "kl_loss", "proto_loss", "dist"], outs[1:]):
        assert rel(o, g[n]) < tol, n


@pytest.mark.parametrize("tag", ["small_idt_passion_1011", "small_idt_passion_b2", "small_idt_plain_1111",
                                 "small_pdt_passion_1101"])
def test_small_model_matches_reference(tag):
    g = gold(f"model_{tag}.npz")
    P, loss, outs, parts = _run_oracle(g, small_weights())
    _check_outputs(g, outs)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    if "rp_mask" in g.files:
        assert np.array_equal(parts["rp_mask"].numpy(), g["rp_mask"])
    loss.backward()
    gn = grad_norms((k, v.grad) for k, v in P.items())
    for k, ref in zip(g["grad_keys"], g["grad_norms"]):
        assert abs(gn[str(k)] - ref) < 2e-4 * ref + 1e-7, (k, gn[str(k)], ref)
    for key in g.files:
        if key.startswith("grad::"):
            if np.abs(g[key]).max() < 1e-6:   # bias in front of an InstanceNorm: gradient is rounding noise around 0
                assert P[key[6:]].grad.abs().max() < 1e-6
                continue
            assert rel(P[key[6:]].grad, g[key]) < 2e-3, key


def test_product_model_tree_matches_reference_weights():
    """state-dict keys/shapes of the product Model equal the reference's (weights_small.npz was saved
    from the reference Model): load_state_dict(strict=True) inside build_product_model asserts it."""
    m = build_product_model(32, small=True)
    sd = m.state_dict()
    w = small_weights()
    assert list(sd.keys()) == list(w.keys())
    assert all(sd[k].shape == w[k].shape for k in w)


def test_full_model_seeded_construction_matches_reference():
    """the full-size fixture stores no weights: the product Model built under torch.manual_seed(1037)
    must consume the RNG exactly like the reference's constructor, then the oracle must reproduce
    the reference's outputs from that state_dict."""
    g = gold("model_full_idt_passion_1111.npz")
    m = build_product_model(32, small=False)
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert len(P) == 279
    P, loss, outs, parts = _run_oracle(g, P)
    _check_outputs(g, outs)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    gn = grad_norms((k, v.grad) for k, v in P.items())
    for k, ref in zip(g["grad_keys"], g["grad_norms"]):
        assert abs(gn[str(k)] - ref) < 5e-4 * ref + 1e-7, (k, gn[str(k)], ref)


def test_oracle_follows_the_reference_training_trajectory():
    """multi-step pin (tools/make_golden_traj.py: the imported reference took K AdamW(amsgrad) steps, train.py:95-96,228-289):
    the oracle's forward + loss assembly + torch AdamW reproduce the loss of the first steps of the 32^3 PASSION trajectory"""
    from _util import signal_volume
    g = gold("traj_passion_32.npz")
    S, steps = int(g["S"]), 6
    m = build_product_model(S, small=False)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    opt = torch.optim.AdamW([{"params": list(P.values()), "lr": float(g["lr"]), "weight_decay": 1e-4}], betas=(0.9, 0.999), eps=1e-08,
                            amsgrad=True)
    x, target = signal_volume(S, g["means"])
    mask = _t(g["mask"])
    for it in range(steps + 1):
        loss, outs, _ = train_step_ref.train_step(P, x, mask, target, 4.0, "idt", bool(g["passion"]))
        assert abs(float(loss.detach()) - g["losses"][it]) < 2e-4 * abs(g["losses"][it]), (it, float(loss.detach()), g["losses"][it])
        opt.zero_grad()
        loss.backward()
        opt.step()
