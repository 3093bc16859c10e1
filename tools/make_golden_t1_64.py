#!/usr/bin/env python
"""The one deviating case of the 15-subset training sweep (mmformer, t1 alone: 1.2 % on the level-4 gradients at 32^3, traced to ONE
element with |x_hat| = 1e-5 on the LeakyReLU kink of a 4^3-voxel tensor) at 64^3, where one element is 1/16384 of that
tensor: the same step of the IMPORTED reference (tools/make_golden_masks.py sweep_grads, one subset) -> tests/golden/t1_alone_64.npz.

    python tools/make_golden_t1_64.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402
import make_golden_masks as mm  # noqa: E402


def main():
    torch.set_num_threads(8)
    S = 64
    mmformer, crit = mg.import_reference()
    m = mg.build(mmformer, S, True)
    x, target = mg.synth(1, S, blobs=True)
    mm.MASKS = [[False, False, True, False]]             # t1 alone
    rec = mm.sweep_grads(m, crit, x, target)
    out = {"S": np.array(S), "masks": np.array(mm.MASKS)}
    out.update({k: v for k, v in rec.items()})
    np.savez_compressed(os.path.join(mg.OUT, "t1_alone_64.npz"), **out)
    print("wrote t1_alone_64.npz: loss", rec["loss"], "gate", rec["rp_mask"])


if __name__ == "__main__":
    main()
