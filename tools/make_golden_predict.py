#!/usr/bin/env python
"""Pin for SURVEY 8(f) N1 (sliding-window inference + Dice), build container only.

code/utils/predict.py cannot be imported here: its header imports nibabel, medpy and scipy.misc, which this image lacks, and no
stand-in modules are written for them.  What CAN be run from it without any of that is `softmax_output_dice_class4`
(predict.py:78-124): the function uses torch only.  This script parses the reference file with `ast`, takes that ONE function
definition out of the tree, compiles it in a namespace that holds the real `torch` / `numpy`, and calls it -- the reference's own
code object, not a copy: nothing of its text is written anywhere.  Golden vectors: Dice (separate + evaluate) on seeded integer
masks including the post-processing branch (< 500 enhancing voxels, :103-106) and an empty class.

The window loop (:182-218) lives inside `test_dice_hd95_softmax`, which also calls medpy's hd95 for every subject: it cannot run
here.  Its pin is therefore one level lower: the IMPORTED reference mmformer.Model (the real forward) driven by
oracle/predict_ref.py's restatement of those lines; the fixture stores the averaged probabilities and the argmax mask, and
tests hold both oracle/predict_ref.py (around oracle/mmformer_ref.py) and the product's utils/predict.py against it.

    python tools/make_golden_predict.py     -> tests/golden/predict.npz
"""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF_FILE = "/root/reference/code/utils/predict.py"


def reference_function(name):
    tree = ast.parse(open(REF_FILE).read(), REF_FILE)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(fn) == 1, name
    mod = ast.Module(body=fn, type_ignores=[])
    ns = {"torch": torch, "np": np}
    exec(compile(mod, REF_FILE, "exec"), ns)
    return ns[name]


def dice_cases():
    g = torch.Generator().manual_seed(11)
    cases = []
    for i, (shape, et) in enumerate([((2, 20, 20, 20), 200), ((2, 20, 20, 20), 1200), ((1, 24, 16, 30), 0), ((3, 12, 12, 12), 499),
                                     ((1, 16, 16, 16), 500)]):
        out = torch.randint(0, 3, shape, generator=g)
        tgt = torch.randint(0, 4, shape, generator=g)
        flat = out.view(-1)
        flat[:et] = 3                                  # exactly `et` enhancing voxels in the whole batch (:103 sums over the batch)
        if i == 2:
            tgt[tgt == 1] = 0                          # a class absent from the target (eps = 1e-8 decides, :80)
        cases.append((out, tgt))
    return cases


def main():
    import make_golden as mg
    dice = reference_function("softmax_output_dice_class4")
    rec = {}
    for i, (out, tgt) in enumerate(dice_cases()):
        sep, ev = dice(out, tgt)
        rec[f"dice{i}_out"], rec[f"dice{i}_tgt"] = out.numpy().astype(np.uint8), tgt.numpy().astype(np.uint8)
        rec[f"dice{i}_sep"], rec[f"dice{i}_ev"] = np.asarray(sep), np.asarray(ev)
    rec["n_dice"] = np.array(len(dice_cases()))
    # ---- the imported reference Model inside the restated window loop
    from oracle import predict_ref
    mmformer, _ = mg.import_reference()
    S = 32
    model = mg.build(mmformer, S, small=True)
    w = np.load(os.path.join(mg.OUT, "weights_small.npz"))
    model.load_state_dict({k: torch.from_numpy(w[k].astype(np.float32)) for k in w.files}, strict=True)
    model.eval()
    model.is_training, model.mask_type, model.use_passion = False, "idt", False
    shape = (48, 40, 56)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, *shape, generator=g)
    tgt = torch.randint(0, 4, (1, *shape), generator=g)
    mask = torch.tensor([[True, False, True, True]])
    with torch.no_grad():
        pred, am = predict_ref.test_volume_softmax(lambda xw, mk: model(xw, mk), x, mask, S)
    sep, ev = dice(am, tgt)
    top2 = pred.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).reshape(-1)
    rec.update({"vol_shape": np.array(shape), "vol_patch": np.array(S), "vol_seed": np.array(5), "vol_mask": mask.numpy(),
                "vol_argmax": am.numpy().astype(np.uint8), "vol_pred_slice": pred[:, :, shape[0] // 2].numpy(),
                "vol_pred_sum": pred.double().sum((2, 3, 4)).numpy(), "vol_sep": np.asarray(sep), "vol_ev": np.asarray(ev),
                "vol_close_idx": np.nonzero((margin < 1e-4).numpy())[0].astype(np.int32)})
    out = os.path.join(mg.OUT, "predict.npz")
    np.savez_compressed(out, **rec)
    print("wrote", out, "dice ev", ev, "close voxels", len(rec["vol_close_idx"]))


if __name__ == "__main__":
    main()
