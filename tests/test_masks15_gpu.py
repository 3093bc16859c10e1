"""GPU: the product networks in INFERENCE mode over all 15 modality subsets (reference predict.py:14-19) against the imported
reference's outputs in tests/golden/masks15.npz: one subset per call in f32 (1e-3, argmax exact outside the reference's own
near-ties), the 15 subsets as ONE batch (per-sample masks inside one launch must not leak between samples), and bf16 storage."""
import numpy as np
import pytest
import torch

from _util import build_m2ftrans, build_product_model, gold, mask_dice, rel, synth
from test_masks15_oracle import STEP_GTOL, STEP_GTOL_ROW, check_step, check_sweep, check_train

pytestmark = pytest.mark.gpu
NAMES = ["mmformer", "rfnet", "m2ftrans"]


def _dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need an MI355X"
    return torch.device("cuda:0")


def _build(name, S):
    if name == "mmformer":
        return build_product_model(S, True)
    if name == "rfnet":
        from test_rfnet_gpu import build_rfnet
        return build_rfnet()
    from test_m2ftrans_oracle import m2f_weights
    m = build_m2ftrans(S, small=True)
    m.load_state_dict(m2f_weights(), strict=True)
    return m


def _setup(name, dtype=torch.float32):
    g = gold("masks15.npz")
    S = int(g["S"])
    dev = _dev()
    m = _build(name, S).to(dev)
    m.is_training, m.compute_dtype = False, dtype
    m.eval()
    x, _ = synth(1, S, blobs=False)
    return g, m, x.to(dev), torch.from_numpy(g["masks"]).to(dev)


@pytest.mark.parametrize("name", NAMES)
def test_inference_over_all_masks_one_at_a_time(name):
    g, m, x, masks = _setup(name)
    with torch.no_grad():
        pred = torch.cat([m(x, masks[i:i + 1]) for i in range(15)])
    assert pred.shape == (15, 4, 32, 32, 32)
    check_sweep(g, name, pred, 1e-3)


@pytest.mark.parametrize("name", NAMES)
def test_inference_over_all_masks_as_one_batch(name):
    g, m, x, masks = _setup(name)
    with torch.no_grad():
        pred = m(x.expand(15, -1, -1, -1, -1).contiguous(), masks)
    check_sweep(g, name, pred, 1e-3)


@pytest.mark.parametrize("name", NAMES)
def test_inference_over_all_masks_bf16(name):
    """bf16 storage on random-noise input (near-uniform probabilities, the hardest case for an argmax; a 2^3-token bottleneck): the
    probabilities stay within 2e-2 of the f32 reference on average per subset and the masks agree on most voxels -- a subset that
    dropped or mixed up a modality would be off by O(1)"""
    g, m, x, masks = _setup(name, torch.bfloat16)
    with torch.no_grad():
        pred = m(x.expand(15, -1, -1, -1, -1).contiguous(), masks).float()
    assert torch.isfinite(pred).all()
    ref = torch.from_numpy(g[name + "_pred_slice"])
    mad = (pred[:, :, 16].cpu() - ref).abs().mean((1, 2, 3))
    am = pred.argmax(1).cpu().numpy().astype(np.uint8)
    agree = (am == g[name + "_argmax"]).reshape(15, -1).mean(1)
    print(f"[masks15 bf16 {name}] mean |dp| per subset max {float(mad.max()):.2e}; mask agreement min {agree.min():.4f}")
    assert float(mad.max()) < 2e-2, mad
    assert agree.min() > 0.9, agree


@pytest.mark.parametrize("mask_type", ["idt", "pdt"])
@pytest.mark.parametrize("name", NAMES)
def test_passion_training_outputs_over_all_masks(name, mask_type):
    """training-mode forward with the PASSION heads (teacher + students, prototypes, distances) for every modality subset against
    the imported reference (masks15_train.npz), f32 1e-3; subsets whose reference output is exactly zero (a single present
    modality has no student to distil) must come out zero"""
    from passion_amd.utils import criterions
    g = gold("masks15_train.npz")
    S = int(g["S"])
    dev = _dev()
    m = _build(name, S).to(dev)
    m.is_training, m.mask_type, m.use_passion, m.compute_dtype = True, mask_type, True, torch.float32
    m.eval()
    x, target = synth(1, S, blobs=True)
    x, target = x.to(dev), target.to(dev)
    with torch.no_grad():
        for row in range(15):
            mask = torch.from_numpy(g["masks"][row:row + 1]).to(dev)
            outs = m(x, mask, target=target, temp=4.0)
            check_train(g, name + ("_pdt" if mask_type == "pdt" else ""), row, outs, criterions.seg_pair_bs(outs[0], target), 1e-3)


@pytest.mark.parametrize("name", NAMES)
def test_training_step_gradients_over_all_masks(name):
    """forward + train.assemble_loss + backward of the product for EVERY modality subset against the imported reference: loss 1e-3,
    the relative-preference gate, the gradient norm of every top-level module -- the encoders of absent modalities must end
    with exactly no gradient, shared decoders with the present ones' only"""
    from _util import grad_norms
    from passion_amd import train as ptrain
    g = gold("masks15_train.npz")
    S = int(g["S"])
    dev = _dev()
    m = _build(name, S).to(dev)
    m.is_training, m.mask_type, m.use_passion, m.compute_dtype = True, "idt", True, torch.float32
    m.eval()
    x, target = synth(1, S, blobs=True)
    x, target = x.to(dev), target.to(dev)
    worst = 0.0
    for row in range(15):
        mask = torch.from_numpy(g["masks"][row:row + 1]).to(dev)
        m.zero_grad(set_to_none=True)
        outs = m(x, mask, target=target, temp=4.0)
        loss = ptrain.assemble_loss(outs, target, mask, "idt", True)
        loss.backward()
        rp = ptrain.relative_preference(outs[5], mask.float(), True).cpu().numpy()
        gn = grad_norms((k, p.grad) for k, p in m.named_parameters())
        w = check_step(g, name, row, loss.detach(), rp, gn, 1e-3, STEP_GTOL_ROW.get((name, row), STEP_GTOL[name]))
        if (name, row) not in STEP_GTOL_ROW:
            worst = max(worst, w)
    print(f"[masks15 step {name}] worst relative gradient-norm error over the subsets held to {STEP_GTOL[name]:g}: {worst:.2e}")


def test_t1_alone_at_64_cubed_has_no_deviation():
    """the 15-subset sweep keeps mmformer / t1 alone with a 2e-2 bound at 32^3 (one LeakyReLU-kink element of a 4^3-voxel tensor,
    STEP_GTOL_ROW above); the same step of the imported reference at 64^3 (tools/make_golden_t1_64.py), where one element is 1 / 16384 of
    that tensor, must hold the ordinary 3e-3: the deviation is a property of the 32^3 fixture, not of a kernel"""
    from _util import grad_norms
    from passion_amd import train as ptrain
    g = gold("t1_alone_64.npz")
    S = int(g["S"])
    dev = _dev()
    # the small-constants model of the fixtures at 64^3: the committed weights (weights_small.npz, built for 32^3) for everything but the
    # position embeddings, whose token count follows the volume -- those as tools/make_golden.py's build() makes them (generator seed + 1,
    # rounded to float16)
    from _util import SMALL_DIMS, FULL_DIMS, small_weights
    from passion_amd.models import mmformer
    try:
        mmformer.patch_size = S // 16
        for k, v in SMALL_DIMS.items():
            setattr(mmformer, k, v)
        torch.manual_seed(1037)
        m = mmformer.Model(num_cls=4)
    finally:
        for k, v in FULL_DIMS.items():
            setattr(mmformer, k, v)
        mmformer.patch_size = 5
    gen = torch.Generator().manual_seed(1038)
    with torch.no_grad():
        for n in ("flair_pos", "t1ce_pos", "t1_pos", "t2_pos"):
            getattr(m, n).copy_((torch.randn(getattr(m, n).shape, generator=gen) * 0.1).half().float())
    res = m.load_state_dict({k: v for k, v in small_weights().items() if not k.endswith("_pos")}, strict=False)
    assert not res.unexpected_keys and all(k.endswith("_pos") for k in res.missing_keys), res
    m = m.to(dev)
    m.is_training, m.mask_type, m.use_passion, m.compute_dtype = True, "idt", True, torch.float32
    m.eval()
    x, target = synth(1, S, blobs=True)
    x, target = x.to(dev), target.to(dev)
    mask = torch.from_numpy(g["masks"][0:1]).to(dev)
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", True)
    loss.backward()
    rp = ptrain.relative_preference(outs[5], mask.float(), True).cpu().numpy()
    gn = grad_norms((k, p.grad) for k, p in m.named_parameters())
    gg = {"mmformer_step_rp_mask": g["rp_mask"], "mmformer_step_loss": g["loss"], "mmformer_step_grad_keys": g["grad_keys"],
          "mmformer_step_grad_norms": g["grad_norms"]}
    worst = check_step(gg, "mmformer", 0, loss.detach(), rp, gn, 1e-3, STEP_GTOL["mmformer"])
    print(f"[t1 alone, 64^3] worst relative gradient-norm error {worst:.2e} (32^3 fixture: 1.2e-2)")
