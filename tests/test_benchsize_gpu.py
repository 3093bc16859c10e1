"""Parity AT THE BENCHMARK SIZE (BASELINE.json configs[1] / configs[2], 4 x 128^3) against fixtures the imported
reference produced at that size (tools/make_golden_128.py: f32, eval(), forward + train.py:228-280 assembly + backward).

  * f32 mode (csrc/conv.hip f32 MFMA = exact fmaf chains): the north_star bar -- outputs, losses, loss < 1e-3 relative,
    gradient norms < 3e-3, gradient tensors < 5e-3; argmax masks compared voxel by voxel: the number of differing
    voxels is REPORTED (printed and asserted against a ceiling), and every one of them must be a tie of the two best
    classes below 1e-5 in the reference's own probabilities.
  * bf16 mode = the BENCHMARKED path (lean bf16 kernels, encoder / decoder_sep side streams on, flash attention),
    dropout off as in the fixture: stated tolerances below, forward and backward.
configs[2]'s fixture carries a mask drawn from brats_split mr2468 and non-unit modal_weight / imb_beta."""
import numpy as np
import pytest
import torch

from _util import build_product_model, gold, grad_norms, rel, synth

pytestmark = pytest.mark.gpu

# bf16 storage / f32 accumulate vs the f32 reference, 128^3, ~130 conv layers deep (measured on MI355X, x ~2 headroom)
# measured (r02, cfg1, marching conv + flash attention): pred max 9.8e-2, rms 2.5e-2, per-class sums 2.0e-3, loss terms 1.6e-4,
# loss 1.1e-4, grad norms 1.4e-2, mask agreement 0.9834 (random-init logits are near-ties almost everywhere: 477 voxels tie
# below 1e-4 even in f32); cfg3 (RFNet 96^3, PASSION): pred max 1.2e-1, rms 2.5e-2, sums 6e-4, loss terms 1.2e-2 (the
# prototype term: a difference of two cosine maps that nearly cancel), loss 1.2e-4, grad norms 4.5e-3, agreement 0.9746
BF16_TOL = {"pred_max": 2e-1, "pred_rms": 5e-2, "pred_sum": 5e-3, "loss_terms": 3e-2, "loss": 2e-3, "grad_norm": 3e-2,
            "mask_agree": 0.97}


def _unpack2(packed, n):
    p = np.asarray(packed)
    out = np.empty((p.size, 4), dtype=np.uint8)
    for j in range(4):
        out[:, j] = (p >> (2 * j)) & 3
    return out.reshape(-1)[:n]


def _state(g, dev):
    from passion_amd import train as ptrain
    st = ptrain.LossState(dev, modal_weight=torch.from_numpy(g["modal_weight"]))
    st.imb_beta = torch.from_numpy(g["imb_beta"]).to(dev)
    return st


def _run(g, dtype, rfnet=False):
    from passion_amd import train as ptrain
    dev = torch.device("cuda:0")
    S = int(g["S"])
    if rfnet:
        from test_rfnet_gpu import build_rfnet
        m = build_rfnet().to(dev)
    else:
        m = build_product_model(S, small=False).to(dev)
    m.mask_type, m.use_passion, m.is_training, m.compute_dtype = "idt", bool(g["passion"]), True, dtype
    x, target = synth(1, S, blobs=bool(g["blobs"]))
    mask = torch.from_numpy(g["mask"]).to(dev)
    x, target = x.to(dev), target.to(dev)
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", m.use_passion, _state(g, dev))
    loss.backward()
    torch.cuda.synchronize()
    return m, outs, loss


def _mask_report(g, pred, tag):
    S = int(g["S"])
    ref = _unpack2(g["argmax_packed"], S ** 3)
    got = pred.argmax(1).reshape(-1).cpu().numpy().astype(np.uint8)
    diff = np.nonzero(ref != got)[0]
    print(f"[{tag}] argmax mask: {len(diff)} of {S ** 3} voxels differ from the reference "
          f"(reference has {len(g['close_idx'])} voxels with top-2 margin < 1e-4, min margin {float(g['margin_min']):.2e})")
    return diff


def _check_f32(g, m, outs, loss, tag):
    S = int(g["S"])
    pred = outs[0]
    assert rel(pred[:, :, S // 2], g["pred_slice"]) < 1e-3
    assert rel(pred.double().sum((2, 3, 4)), g["pred_sum"]) < 1e-3
    for n, o in zip(["prm_loss", "sep_loss", "kl_loss", "proto_loss", "dist"], outs[1:]):
        assert rel(o, g[n]) < 1e-3, (n, o, g[n])
    assert abs(float(loss) - float(g["loss"])) < 1e-3 * abs(float(g["loss"])), (float(loss), float(g["loss"]))
    diff = _mask_report(g, pred, tag)
    # bit-exact except ties: each differing voxel must be one the REFERENCE itself decides by less than 1e-4 (close_idx)
    # and the product by less than 1e-5
    assert len(diff) <= 16, f"{len(diff)} voxels differ"
    if len(diff):
        assert np.isin(diff, g["close_idx"]).all(), "mask differs at a voxel the reference decides by more than 1e-4"
        top2 = pred.detach().reshape(4, -1)[:, torch.from_numpy(diff).to(pred.device)].topk(2, dim=0).values
        assert float((top2[0] - top2[1]).max()) < 1e-5
    gn = grad_norms((k, p.grad) for k, p in m.named_parameters())
    for k, ref in zip(g["grad_keys"], g["grad_norms"]):
        assert abs(gn[str(k)] - ref) < 3e-3 * ref + 1e-6, (k, gn[str(k)], ref)
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad::") and np.abs(g[key]).max() > 1e-6:
            assert rel(params[key[6:]].grad, g[key]) < 5e-3, (key, rel(params[key[6:]].grad, g[key]))


def _check_bf16(g, m, outs, loss, tag):
    S = int(g["S"])
    pred = outs[0]
    t = BF16_TOL
    ref_slice = torch.from_numpy(g["pred_slice"]).double()
    obs = {"pred_max": rel(pred[:, :, S // 2], g["pred_slice"]),
           "pred_rms": float((pred[:, :, S // 2].double().cpu() - ref_slice).pow(2).mean().sqrt() / ref_slice.pow(2).mean().sqrt()),
           "pred_sum": rel(pred.double().sum((2, 3, 4)), g["pred_sum"])}
    obs["loss_terms"] = max(rel(o, g[n]) for n, o in zip(["prm_loss", "sep_loss", "kl_loss", "proto_loss"], outs[1:5]))
    obs["loss"] = abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))
    gn = grad_norms((k, p.grad) for k, p in m.named_parameters())
    obs["grad_norm"] = max(abs(gn[str(k)] - ref) / (ref + 1e-12) for k, ref in zip(g["grad_keys"], g["grad_norms"]) if ref > 1e-6)
    diff = _mask_report(g, pred, tag)
    agree = 1.0 - len(diff) / S ** 3
    print(f"[{tag}] bf16 vs f32 reference: " + ", ".join(f"{k} {v:.3e}" for k, v in obs.items()) + f", mask agreement {agree:.5f}")
    for k, v in obs.items():
        assert v < t[k], (k, v, t[k])
    assert agree > t["mask_agree"]
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_config1_mmformer128_plain_f32_vs_reference():
    g = gold("model_full128_idt_plain_1111.npz")
    _check_f32(g, *_run(g, torch.float32), "cfg1 f32")


def test_config2_mmformer128_passion_split_f32_vs_reference():
    g = gold("model_full128_idt_passion_split.npz")
    assert not np.allclose(g["modal_weight"], 1) and not np.allclose(g["imb_beta"], 1) and not g["mask"].all()
    m, outs, loss = _run(g, torch.float32)
    _check_f32(g, m, outs, loss, "cfg2 f32")
    from passion_amd import train as ptrain
    mk = torch.from_numpy(g["mask"]).float().to(outs[5].device)
    assert torch.equal(ptrain.relative_preference(outs[5], mk, True).cpu(), torch.from_numpy(g["rp_mask"]))


def test_config1_mmformer128_plain_bf16_benchmarked_path():
    from passion_amd.models import mmformer
    assert mmformer.ENC_STREAMS, "the benchmarked path runs the encoders on side streams"
    g = gold("model_full128_idt_plain_1111.npz")
    _check_bf16(g, *_run(g, torch.bfloat16), "cfg1 bf16")


def test_config2_mmformer128_passion_split_bf16_benchmarked_path():
    g = gold("model_full128_idt_passion_split.npz")
    _check_bf16(g, *_run(g, torch.bfloat16), "cfg2 bf16")


def test_config3_rfnet96_passion_split_f32_vs_reference():
    g = gold("rfnet_96_idt_passion_split.npz")
    _check_f32(g, *_run(g, torch.float32, rfnet=True), "cfg3 f32 (96^3)")


def test_config3_rfnet96_passion_split_bf16():
    g = gold("rfnet_96_idt_passion_split.npz")
    _check_bf16(g, *_run(g, torch.bfloat16, rfnet=True), "cfg3 bf16 (96^3)")


def test_reference_default_crop_80_f32_vs_reference():
    """the reference's OWN configuration (patch_size 5: 80^3 crops, options.py:50; levels 80 / 40 / 20 / 10 / 5 are ragged against the
    4 x 8 x 8 bricks and 80 % 16 != 0 keeps the marching kernels out: the lean / general kernel paths), PASSION step with the
    brats_split mask, against a fixture of the imported reference (tools/make_golden_128.py crop80)"""
    g = gold("model_full80_idt_passion_split.npz")
    assert int(g["S"]) == 80
    _check_f32(g, *_run(g, torch.float32), "crop 80 f32")


def test_reference_default_crop_80_bf16():
    g = gold("model_full80_idt_passion_split.npz")
    _check_bf16(g, *_run(g, torch.bfloat16), "crop 80 bf16")
