#!/usr/bin/env python
"""K-step golden trajectories for the other two backbones: the IMPORTED reference's rfnet.Model / m2ftrans.Model
(/root/reference/code, build container only) take K AdamW(amsgrad) steps from the committed float16-representable start weights
(tests/golden/weights_rfnet.npz / weights_m2f.npz: what tools/make_golden.py / make_golden_m2f.py build from seed 1037) on the
label-carrying synthetic volume of tools/make_golden_traj.py, with the loss assembly of train.py:228-280 (imb_beta = modal_weight = 1).
Committed per fixture: the loss of every step, the voxel accuracy, the final argmax mask (2-bit packed), the voxels the reference
decides by small margins, PASSION's relative-preference gates and signed distances.

    python tools/make_golden_traj_backbones.py rfnet 32 12 passion      -> tests/golden/traj_rfnet_passion_32.npz
    python tools/make_golden_traj_backbones.py m2ftrans 32 12 passion   -> tests/golden/traj_m2ftrans_passion_32.npz
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402
import make_golden_traj as mt  # noqa: E402


def main():
    backbone, S, K, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    passion = mode == "passion"
    torch.set_num_threads(int(os.environ.get("TRAJ_THREADS", "8")))
    _, crit = mg.import_reference()
    if backbone == "rfnet":
        from models import rfnet
        torch.manual_seed(1037)
        m = rfnet.Model(num_cls=4)
        with torch.no_grad():
            for p_ in m.parameters():
                p_.copy_(p_.half().float())
        m.eval()
        wfile = "weights_rfnet.npz"
    else:
        import make_golden_m2f as m2g
        from models import m2ftrans
        m = m2g.build(m2ftrans, S)
        wfile = "weights_m2f.npz"
    w0 = np.load(os.path.join(mg.OUT, wfile))
    for k, v in m.state_dict().items():              # the committed start weights are what this run starts from
        assert np.array_equal(v.numpy().astype(np.float16), w0[k]), k
    m.mask_type, m.use_passion, m.is_training = "idt", passion, True
    x, target = mt.signal_volume(S)
    mask = torch.tensor([[True, True, False, True]])
    mk = mask.float()
    opt = torch.optim.AdamW([{"params": m.parameters(), "lr": mt.LR, "weight_decay": 1e-4}], betas=(0.9, 0.999), eps=1e-08, amsgrad=True)
    losses, accs, rps, rpv = [], [], [], []
    lab = target.argmax(1)
    for it in range(K + 1):
        t0 = time.time()
        outs = m(x, mask, target=target, temp=4.0)
        pred = outs[0]
        fuse = crit.softmax_weighted_loss_bs(pred, target, num_cls=4) + crit.dice_loss_bs(pred, target, num_cls=4)
        loss = fuse.sum() + outs[1].sum()
        if passion:
            sep_m, kl_m, proto_m = (outs[2] * mk).sum(0), (outs[3] * mk).sum(0), (outs[4] * mk).sum(0)
            d = outs[5][0]
            rp = mk[0] * (d / (d.sum() / mk[0].sum()) - 1)
            rp_mask = (rp > 0).float()
            rps.append(rp_mask.numpy().copy())
            rpv.append(rp.detach().numpy().copy())
            loss = loss + (rp_mask * sep_m).sum() + 0.5 * kl_m.sum() + 0.1 * (rp_mask * proto_m).sum()
        else:
            loss = loss + (outs[2] * mk).sum()
        losses.append(float(loss))
        accs.append(float((pred.argmax(1) == lab).float().mean()))
        print(f"step {it}: loss {losses[-1]:.6f} acc {accs[-1]:.4f} ({time.time() - t0:.1f} s fwd)", flush=True)
        if it == K:
            break
        opt.zero_grad()
        loss.backward()
        opt.step()
    pred = pred.detach()
    top2 = pred.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).reshape(-1)
    am = pred.argmax(1).reshape(-1).numpy().astype(np.uint8)
    rec = {"backbone": backbone, "S": S, "K": K, "passion": passion, "lr": mt.LR, "mask": mask.numpy(), "means": np.array(mt.MEANS, np.float32),
           "losses": np.array(losses), "accs": np.array(accs), "argmax_packed": mt.pack2(am),
           "close3_idx": np.nonzero((margin < 1e-1).numpy())[0].astype(np.int32), "margin_min": float(margin.min()),
           "pred_slice": pred[:, :, S // 2].numpy(), "label_packed": mt.pack2(lab.reshape(-1).numpy())}
    if passion:
        rec["rp_masks"] = np.stack(rps)
        rec["rp_vals"] = np.stack(rpv)
    out = os.path.join(mg.OUT, f"traj_{backbone}_{mode}_{S}.npz")
    np.savez_compressed(out, **rec)
    print("wrote", out, "final loss", losses[-1], "acc", accs[-1], "size", os.path.getsize(out))


if __name__ == "__main__":
    main()
