#!/usr/bin/env python
"""Golden fixture of the INFERENCE path over all 15 modality subsets (the "erasure" cases of this domain: reference
predict.py:14-19 evaluates every checkpoint on exactly these masks) from the IMPORTED reference (/root/reference/code, build
container only; same shims and builders as tools/make_golden.py / make_golden_m2f.py: small-constants mmformer / m2ftrans and the
full RFNet with float16-representable weights, identical to the committed weights_small / weights_m2f / weights_rfnet fixtures --
asserted below).  model.is_training = False, model.eval(), torch.no_grad(): forward returns the fused prediction only.

    python tools/make_golden_masks.py      -> tests/golden/masks15.npz, masks15_train.npz (the PASSION training outputs per subset)

Per model: argmax (15, S, S, S) uint8, the mid-depth slice of the probabilities (15, 4, S, S), their per-class sums (15, 4) and
`tie` (15, S^3 / 8 packed bits): voxels whose top-2 margin is below 1e-3 in the reference, where an argmax may legitimately flip."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402

S = 32
MASKS = [[bool(k >> (3 - i) & 1) for i in range(4)] for k in range(1, 16)]      # 0001 .. 1111 over (flair, t1ce, t1, t2)


def same_as_fixture(model, name):
    w = np.load(os.path.join(mg.OUT, name))
    sd = model.state_dict()
    assert set(w.files) == set(sd.keys()), name
    for k in w.files:
        assert np.array_equal(w[k].astype(np.float32), sd[k].numpy()), (name, k)


def sweep(model, x):
    model.is_training = False
    model.eval()
    am, sl, sm, tie = [], [], [], []
    with torch.no_grad():
        for mk in MASKS:
            pred = model(x, torch.tensor([mk]))
            assert pred.shape == (1, 4, S, S, S)
            am.append(pred.argmax(1)[0].numpy().astype(np.uint8))
            sl.append(pred[0, :, S // 2].numpy())
            sm.append(pred[0].double().sum((1, 2, 3)).numpy())
            top2 = pred[0].topk(2, dim=0).values
            tie.append(np.packbits(((top2[0] - top2[1]) < 1e-3).numpy().reshape(-1)))
    return {"argmax": np.stack(am), "pred_slice": np.stack(sl), "pred_sum": np.stack(sm), "tie": np.stack(tie)}


def sweep_train(model, crit, x, target, mask_type="idt"):
    """training-mode forward with the PASSION heads (mask_type idt, temp 4, dropout off) per subset: the fused prediction's loss pair
    and the five loss / distance outputs of Model.forward (reference mmformer.py:610-659 and its rfnet / m2ftrans twins)"""
    model.is_training, model.mask_type, model.use_passion = True, mask_type, True
    model.eval()
    rec = {k: [] for k in ("fuse_loss_bs", "prm_loss", "sep_loss", "kl_loss", "proto_loss", "dist", "pred_sum")}
    with torch.no_grad():
        for mk in MASKS:
            outs = model(x, torch.tensor([mk]), target=target, temp=4.0)
            pred = outs[0]
            rec["fuse_loss_bs"].append((crit.softmax_weighted_loss_bs(pred, target, num_cls=4) + crit.dice_loss_bs(pred, target, num_cls=4)).numpy())
            rec["pred_sum"].append(pred.double().sum((2, 3, 4)).numpy())
            for n, o in zip(["prm_loss", "sep_loss", "kl_loss", "proto_loss", "dist"], outs[1:]):
                rec[n].append(o.detach().numpy())
    return {k: np.stack(v) for k, v in rec.items()}


def sweep_grads(model, crit, x, target):
    """one training step's loss (train.py:228-280 with imb_beta = modal_weight = 1, as tools/make_golden.py assembles it) and the
    gradient norm of every top-level module, per subset: a masked modality's encoder must end up with a zero gradient, the shared
    decoders with the gradient of the present ones only"""
    model.is_training, model.mask_type, model.use_passion = True, "idt", True
    model.eval()
    losses, rps, norms, keys = [], [], [], None
    for mk in MASKS:
        mask_t = torch.tensor([mk])
        outs = model(x, mask_t, target=target, temp=4.0)
        fuse = crit.softmax_weighted_loss_bs(outs[0], target, num_cls=4) + crit.dice_loss_bs(outs[0], target, num_cls=4)
        m_ = mask_t.float()
        sep_m, kl_m, proto_m = (outs[2] * m_).sum(0), (outs[3] * m_).sum(0), (outs[4] * m_).sum(0)
        rp = m_[0] * (outs[5][0] / (outs[5][0].sum() / m_[0].sum()) - 1)
        rp_mask = (rp > 0).float()
        loss = fuse.sum() + outs[1].sum() + (rp_mask * sep_m).sum() + 0.5 * kl_m.sum() + 0.1 * (rp_mask * proto_m).sum()
        model.zero_grad()
        loss.backward()
        gn = mg.grad_norms(model)
        keys = sorted(gn) if keys is None else keys
        assert sorted(gn) == keys
        losses.append(float(loss)); rps.append(rp_mask.numpy().copy()); norms.append([gn[k] for k in keys])
    return {"loss": np.array(losses), "rp_mask": np.stack(rps), "grad_norms": np.array(norms), "grad_keys": np.array(keys)}


def main():
    torch.set_num_threads(8)
    mmformer, crit = mg.import_reference()
    from models import m2ftrans, rfnet
    import make_golden_m2f as m2g
    x, _ = mg.synth(1, S, blobs=False)
    out = {"S": np.array(S), "masks": np.array(MASKS)}
    m = mg.build(mmformer, S, True)
    same_as_fixture(m, "weights_small.npz")
    for k, v in sweep(m, x).items():
        out["mmformer_" + k] = v
    torch.manual_seed(1037)
    r = rfnet.Model(num_cls=4)
    with torch.no_grad():
        for p_ in r.parameters():
            p_.copy_(p_.half().float())
    same_as_fixture(r, "weights_rfnet.npz")
    for k, v in sweep(r, x).items():
        out["rfnet_" + k] = v
    f = m2g.build(m2ftrans, S)
    same_as_fixture(f, "weights_m2f.npz")
    for k, v in sweep(f, x).items():
        out["m2ftrans_" + k] = v
    np.savez_compressed(os.path.join(mg.OUT, "masks15.npz"), **out)
    xb, tb = mg.synth(1, S, blobs=True)
    tr = {"S": np.array(S), "masks": np.array(MASKS)}
    for name, mod in (("mmformer", m), ("rfnet", r), ("m2ftrans", f)):
        for k, v in sweep_train(mod, crit, xb, tb).items():
            tr[f"{name}_{k}"] = v
        for k, v in sweep_train(mod, crit, xb, tb, "pdt").items():          # mask_type 'pdt' (train.py:234-237): its own branch in forward
            tr[f"{name}_pdt_{k}"] = v
    for name, mod in (("mmformer", m), ("rfnet", r), ("m2ftrans", f)):
        for k, v in sweep_grads(mod, crit, xb, tb).items():
            tr[f"{name}_step_{k}"] = v
    np.savez_compressed(os.path.join(mg.OUT, "masks15_train.npz"), **tr)
    print("masks15_train.npz:", {k: v.shape for k, v in tr.items() if k.startswith("mmformer_")})
    for name in ("mmformer", "rfnet", "m2ftrans"):
        print(name, "ties per mask:", [int(np.unpackbits(t).sum()) for t in out[name + "_tie"]])


if __name__ == "__main__":
    main()
