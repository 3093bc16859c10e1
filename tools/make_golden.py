#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by IMPORTING the real reference
(/root/reference/code) in the build container.  The reference source never travels: only the
inputs' seeds, small weights and outputs written here are committed.

Shims applied to the reference at import time (oracle side only):
  * torch.Tensor.cuda -> identity   (code/utils/criterions.py:153 hard-codes .cuda())
  * mmformer.patch_size = S // 16   (module constant 5 only fits 80^3 crops, mmformer.py:21)
  * model.eval() (dropout off) with model.is_training = True
  * "small" variants shrink the module constants basic_dims / transformer_basic_dims / mlp_dim to
    4 / 64 / 128 and round the weights to float16 values, so that the whole state_dict can be stored
    once (weights_small.npz) and parity does not depend on RNG order.

Usage: python tools/make_golden.py            (writes tests/golden/*.npz, ~ a minute on 8 cores)
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/code"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    warnings.filterwarnings("ignore")
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models import mmformer  # noqa
    from utils import criterions  # noqa
    return mmformer, criterions


def synth(B, S, seed=7, blobs=False, radii=(0.35, 0.25, 0.12)):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, S, S, S, generator=g)
    if blobs:
        zz, yy, xx = torch.meshgrid(*[torch.arange(S, dtype=torch.float32)] * 3, indexing="ij")
        r = ((zz - S / 2) ** 2 + (yy - S / 2) ** 2 + (xx - S / 2) ** 2).sqrt()
        lab = torch.zeros(B, S, S, S, dtype=torch.long)
        lab[:, r < S * radii[0]] = 2
        lab[:, r < S * radii[1]] = 1
        lab[:, r < S * radii[2]] = 3
    else:
        lab = torch.randint(0, 4, (B, S, S, S), generator=g)
    target = F.one_hot(lab, 4).permute(0, 4, 1, 2, 3).double()   # float64 one-hot (datasets_nii.py:151-153)
    return x, target


def build(mmformer, S, small, seed=1037):
    mmformer.patch_size = S // 16
    if small:
        mmformer.basic_dims, mmformer.transformer_basic_dims, mmformer.mlp_dim = 4, 64, 128
    else:
        mmformer.basic_dims, mmformer.transformer_basic_dims, mmformer.mlp_dim = 8, 512, 4096
    torch.manual_seed(seed)
    m = mmformer.Model(num_cls=4)
    # pos embeddings are zero-initialised in the reference; give them values so they are exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n in ("flair_pos", "t1ce_pos", "t1_pos", "t2_pos"):
            getattr(m, n).copy_(torch.randn(getattr(m, n).shape, generator=g) * 0.1)
        if small:   # weights are stored as float16 in the fixture: make them exactly representable
            for p_ in m.parameters():
                p_.copy_(p_.half().float())
    m.eval()
    return m


def grad_norms(model):
    out = {}
    for name, p in model.named_parameters():
        top = name.split(".")[0]
        if p.grad is not None:
            out[top] = out.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    return {k: v ** 0.5 for k, v in out.items()}


def model_case(mmformer, criterions, tag, S, B, mask, mask_type, passion, small, blobs, temp=4.0, with_grads=True):
    model = build(mmformer, S, small)
    model.mask_type, model.use_passion, model.is_training = mask_type, passion, True
    x, target = synth(B, S, blobs=blobs)
    mask_t = torch.tensor(mask, dtype=torch.bool).view(B, 4)
    outs = model(x, mask_t, target=target, temp=temp)
    fuse_pred = outs[0]
    fuse_loss_bs = criterions.softmax_weighted_loss_bs(fuse_pred, target, num_cls=4) + \
        criterions.dice_loss_bs(fuse_pred, target, num_cls=4)
    rec = {"S": S, "B": B, "mask": np.array(mask, dtype=bool).reshape(B, 4), "temp": temp,
           "mask_type": mask_type, "passion": passion, "small": small, "blobs": blobs,
           "fuse_loss_bs": fuse_loss_bs.detach().numpy(), "argmax": fuse_pred.argmax(1).numpy().astype(np.uint8)}
    rec["pred_slice"] = fuse_pred[:, :, S // 2].detach().numpy()           # (B,4,S,S) central depth slice
    rec["pred_sum"] = fuse_pred.detach().double().sum((2, 3, 4)).numpy()
    names = ["prm_loss", "sep_loss", "kl_loss", "proto_loss", "dist"]
    for n, o in zip(names, outs[1:]):
        rec[n] = o.detach().numpy()
    if with_grads:
        # train.py:228-280 loss assembly with imb_beta = modal_weight = 1 and rp_mask from dist
        loss = fuse_loss_bs.sum() + outs[1].sum()
        mk = mask_t.float() if mask_type != "pdt" else torch.ones(B, 4)     # pdt branch sums unmasked (train.py:234-237)
        if passion:
            sep_m, kl_m, proto_m = (outs[2] * mk).sum(0), (outs[3] * mk).sum(0), (outs[4] * mk).sum(0)
            dist_bs = outs[5]
            rp = torch.zeros(4)
            for b in range(B):   # train.py:239-241 (pdt: mean over 4) / :265-267 (idt: mean over present modalities)
                rp += mk[b] * (dist_bs[b] / (dist_bs[b].sum() / mk[b].sum()) - 1)
            rp_mask = (rp > 0).float()
            rec["rp_mask"] = rp_mask.numpy()
            loss = loss + (rp_mask * sep_m).sum() + 0.5 * kl_m.sum() + 0.1 * (rp_mask * proto_m).sum()
        else:
            loss = loss + (outs[2] * mk).sum()
        rec["loss"] = float(loss)
        model.zero_grad()
        loss.backward()
        gn = grad_norms(model)
        rec["grad_keys"] = np.array(sorted(gn))
        rec["grad_norms"] = np.array([gn[k] for k in sorted(gn)])
        # a few full gradient tensors for element-wise checks
        sd = dict(model.named_parameters())
        for key in ("flair_encoder.e1_c1.weight", "t2_encoder.e3_c2.conv.weight", "decoder_fuse.RFM1.fusion_layer.0.conv.weight",
                    "decoder_fuse.d1_c2.conv.bias", "decoder_sep.d2_c1.conv.weight", "multimodal_decode_conv.bias",
                    "flair_transformer.cross_attention_list.0.fn.fn.qkv.weight" if small else "t1_encode_conv.bias",
                    "t1_pos"):
            if sd[key].grad is not None:
                rec["grad::" + key] = sd[key].grad.numpy()
    if small:
        wpath = os.path.join(OUT, "weights_small.npz")
        if not os.path.exists(wpath):
            np.savez_compressed(wpath, **{k: v.numpy().astype(np.float16) for k, v in model.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, f"model_{tag}.npz"), **rec)
    print("wrote", tag, {k: (v.tolist() if hasattr(v, 'tolist') and np.size(v) <= 8 else '...') for k, v in rec.items()
                         if k in ("fuse_loss_bs", "prm_loss", "sep_loss", "kl_loss", "proto_loss", "dist", "loss")})


def loss_cases(mmformer, criterions):
    g = torch.Generator().manual_seed(21)
    B, S = 2, 16
    rec = {}
    x, target = synth(B, S, seed=3, blobs=False)
    _, target_b = synth(B, S, seed=3, blobs=True)
    target_b[1, 0] += target_b[1, 3]    # sample 1 loses class 3 -> proto skips it for the whole batch (criterions.py:157)
    target_b[1, 3] = 0
    logits = torch.randn(B, 4, S, S, S, generator=g) * 5
    logits_t = torch.randn(B, 4, S, S, S, generator=g) * 5
    probs = torch.softmax(logits, 1)
    rec["logits"], rec["logits_t"] = logits.numpy(), logits_t.numpy()
    rec["target"] = target.numpy().astype(np.uint8)
    rec["target_b"] = target_b.numpy().astype(np.uint8)
    for tn, t in (("rand", target), ("blob", target_b)):
        rec[f"wce_{tn}"] = criterions.softmax_weighted_loss_bs(probs, t, num_cls=4).numpy()
        rec[f"dice_{tn}"] = criterions.dice_loss_bs(probs, t, num_cls=4).numpy()
    rec["kl_T4"] = criterions.temp_kl_loss_bs(logits, logits_t, target, num_cls=4, temp=4.0).numpy()
    rec["kl_T1"] = criterions.temp_kl_loss_bs(logits, logits_t, target, num_cls=4, temp=1.0).numpy()
    for k in (2, 4, 8):
        s = S // k
        lo = torch.randn(B, 4, s, s, s, generator=g) * 3
        lo_t = torch.randn(B, 4, s, s, s, generator=g) * 3
        up = torch.nn.Upsample(scale_factor=k, mode="trilinear", align_corners=True)
        rec[f"lo{k}"], rec[f"lo{k}_t"] = lo.numpy(), lo_t.numpy()
        rec[f"wce_up{k}"] = criterions.softmax_weighted_loss_bs(torch.softmax(lo, 1), target, num_cls=4, up_op=up).numpy()
        rec[f"dice_up{k}"] = criterions.dice_loss_bs(torch.softmax(lo, 1), target, num_cls=4, up_op=up).numpy()
        rec[f"kl_up{k}"] = criterions.temp_kl_loss_bs(lo, lo_t, target, num_cls=4, temp=4.0, up_op=up).numpy()
    fs = torch.randn(B, 8, S, S, S, generator=g)
    ft = torch.randn(B, 8, S, S, S, generator=g)
    rec["fs"], rec["ft"] = fs.numpy(), ft.numpy()
    for tn, t in (("rand", target), ("blob", target_b)):
        pl, d = criterions.prototype_passion_loss_bs(fs, ft, t, logits, logits_t, num_cls=4, temp=4.0)
        rec[f"proto_{tn}"], rec[f"protodist_{tn}"] = pl.numpy(), d.numpy()
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **rec)
    print("wrote losses")


def rfnet_case(rfnet, criterions, tag, S, B, mask, mask_type, passion, blobs, temp=4.0):
    """rfnet.Model (reference code/models/rfnet.py) with float16-representable weights stored once"""
    torch.manual_seed(1037)
    model = rfnet.Model(num_cls=4)
    with torch.no_grad():
        for p_ in model.parameters():
            p_.copy_(p_.half().float())
    model.eval()
    model.mask_type, model.use_passion, model.is_training = mask_type, passion, True
    x, target = synth(B, S, blobs=blobs)
    mask_t = torch.tensor(mask, dtype=torch.bool).view(B, 4)
    outs = model(x, mask_t, target=target, temp=temp)
    fuse_pred = outs[0]
    fuse_loss_bs = criterions.softmax_weighted_loss_bs(fuse_pred, target, num_cls=4) + \
        criterions.dice_loss_bs(fuse_pred, target, num_cls=4)
    rec = {"S": S, "B": B, "mask": np.array(mask, dtype=bool).reshape(B, 4), "temp": temp, "mask_type": mask_type,
           "passion": passion, "small": False, "blobs": blobs, "fuse_loss_bs": fuse_loss_bs.detach().numpy(),
           "argmax": fuse_pred.argmax(1).numpy().astype(np.uint8), "pred_slice": fuse_pred[:, :, S // 2].detach().numpy(),
           "pred_sum": fuse_pred.detach().double().sum((2, 3, 4)).numpy()}
    for n, o in zip(["prm_loss", "sep_loss", "kl_loss", "proto_loss", "dist"], outs[1:]):
        rec[n] = o.detach().numpy()
    loss = fuse_loss_bs.sum() + outs[1].sum()
    mk = mask_t.float() if mask_type != "pdt" else torch.ones(B, 4)
    if passion:
        sep_m, kl_m, proto_m = (outs[2] * mk).sum(0), (outs[3] * mk).sum(0), (outs[4] * mk).sum(0)
        rp = torch.zeros(4)
        for b in range(B):
            rp += mk[b] * (outs[5][b] / (outs[5][b].sum() / mk[b].sum()) - 1)
        rp_mask = (rp > 0).float()
        rec["rp_mask"] = rp_mask.numpy()
        loss = loss + (rp_mask * sep_m).sum() + 0.5 * kl_m.sum() + 0.1 * (rp_mask * proto_m).sum()
    else:
        loss = loss + (outs[2] * mk).sum()
    rec["loss"] = float(loss)
    model.zero_grad()
    loss.backward()
    gn = grad_norms(model)
    rec["grad_keys"] = np.array(sorted(gn))
    rec["grad_norms"] = np.array([gn[k] for k in sorted(gn)])
    sd = dict(model.named_parameters())
    for key in ("flair_encoder.e1_c1.conv.weight", "t1_encoder.e3_c2.conv.weight", "decoder_fuse.RFM1.modal_fusion.2.weight_layer.0.weight",
                "decoder_fuse.prm_generator1.embedding_layer.1.conv.weight", "decoder_fuse.RFM3.region_fusion.fusion_layer.0.conv.weight",
                "decoder_fuse.RFM2.short_cut.1.conv.weight", "decoder_fuse.prm_generator4.prm_layer.1.bias", "decoder_sep.d2_c1.conv.weight"):
        if sd[key].grad is not None:
            rec["grad::" + key] = sd[key].grad.numpy()
    wpath = os.path.join(OUT, "weights_rfnet.npz")
    if not os.path.exists(wpath):
        np.savez_compressed(wpath, **{k: v.numpy().astype(np.float16) for k, v in model.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, f"rfnet_{tag}.npz"), **rec)
    print("wrote rfnet", tag, rec["loss"])


def aug_cases():
    """input pipeline: the reference's own train transform chain (options.py:50) run on small synthetic
    volumes under fixed seeds; stores inputs' seeds, the sampled parameters and the outputs
    (x float32 (C,H,W,D), one-hot float64 (4,H,W,D) as datasets_nii.py:143-153 builds them)."""
    import collections
    import collections.abc
    import random
    collections.Sequence = collections.abc.Sequence          # transforms.py:28 predates python 3.10
    sys.path.insert(0, REF)
    from data import transforms as T
    out = {}
    for ci, (shape, size, seed) in enumerate([((40, 36, 30), (24, 24, 24), 3), ((33, 41, 37), (32, 24, 16), 11),
                                              ((24, 24, 24), (24, 24, 24), 5), ((48, 40, 44), (16, 32, 24), 29)]):
        rs = np.random.RandomState(1000 + seed)
        vol = rs.randn(*shape, 4).astype(np.float32)
        seg = rs.randint(0, 4, shape).astype(np.uint8)
        random.seed(seed); np.random.seed(seed)
        tf = T.Compose([T.RandCrop3D(size), T.RandomRotion(10), T.RandomIntensityChange((0.1, 0.1)), T.RandomFlip(0),
                        T.NumpyType((np.float32, np.int64))])
        x, y = tf([vol[None].copy(), seg[None].copy()])
        x = np.ascontiguousarray(x.transpose(0, 4, 1, 2, 3))
        _, H, W, Z = y.shape
        yo = np.eye(4)[y.reshape(-1)].reshape(1, H, W, Z, -1)
        yo = np.ascontiguousarray(yo.transpose(0, 4, 1, 2, 3))
        out[f"c{ci}_shape"], out[f"c{ci}_size"], out[f"c{ci}_seed"] = np.array(shape), np.array(size), np.array(seed)
        out[f"c{ci}_x"], out[f"c{ci}_label"] = x[0], yo[0].argmax(0).astype(np.uint8)
        out[f"c{ci}_onehot_sum"] = yo[0].sum((1, 2, 3))
        assert (np.eye(4)[out[f"c{ci}_label"]].transpose(3, 0, 1, 2) == yo[0]).all()
    out["ncases"] = np.array(4)
    np.savez_compressed(os.path.join(OUT, "augment.npz"), **out)
    print("augment.npz written")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "aug":        # only the input-pipeline fixtures
        aug_cases()
        return
    mmformer, criterions = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "rfnet":      # only (re)generate the RFNet fixtures
        from models import rfnet
        rfnet_case(rfnet, criterions, "idt_passion_1011", 32, 1, [[1, 0, 1, 1]], "idt", True, True)
        rfnet_case(rfnet, criterions, "idt_passion_b2", 32, 2, [[1, 1, 1, 1], [0, 0, 1, 0]], "idt", True, True)
        rfnet_case(rfnet, criterions, "idt_plain_1111", 32, 1, [[1, 1, 1, 1]], "idt", False, False)
        rfnet_case(rfnet, criterions, "pdt_passion_1101", 32, 1, [[1, 1, 0, 1]], "pdt", True, True)
        return
    loss_cases(mmformer, criterions)
    # small-constants models: full state_dict stored -> independent of RNG order
    model_case(mmformer, criterions, "small_idt_passion_1011", 32, 1, [[1, 0, 1, 1]], "idt", True, True, True)
    model_case(mmformer, criterions, "small_idt_passion_b2", 32, 2, [[1, 1, 1, 1], [0, 0, 1, 0]], "idt", True, True, True)
    model_case(mmformer, criterions, "small_idt_plain_1111", 32, 1, [[1, 1, 1, 1]], "idt", False, True, False)
    model_case(mmformer, criterions, "small_pdt_passion_1101", 32, 1, [[1, 1, 0, 1]], "pdt", True, True, True)
    # full-size model (35.6 M parameters): weights reproduced from torch.manual_seed(1037) construction order
    model_case(mmformer, criterions, "full_idt_passion_1111", 32, 1, [[1, 1, 1, 1]], "idt", True, False, True)
    model_case(mmformer, criterions, "full_idt_plain_64", 64, 1, [[1, 1, 1, 1]], "idt", False, False, True, with_grads=False)
    from models import rfnet
    rfnet_case(rfnet, criterions, "idt_passion_1011", 32, 1, [[1, 0, 1, 1]], "idt", True, True)
    rfnet_case(rfnet, criterions, "idt_passion_b2", 32, 2, [[1, 1, 1, 1], [0, 0, 1, 0]], "idt", True, True)
    rfnet_case(rfnet, criterions, "idt_plain_1111", 32, 1, [[1, 1, 1, 1]], "idt", False, False)
    rfnet_case(rfnet, criterions, "pdt_passion_1101", 32, 1, [[1, 1, 0, 1]], "pdt", True, True)
    aug_cases()


if __name__ == "__main__":
    main()
