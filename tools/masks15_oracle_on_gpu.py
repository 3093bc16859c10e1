"""conditioning probe: the ORACLE (plain torch ops) run on the GPU in f32 and f64 for one masks15_train.npz row -- same algorithm, other
reduction orders than the CPU run the fixture came from.  If its f32 gradient norms move by as much as the product's differ from the
fixture, the difference is rounding, not a wrong gradient."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from _util import gold, grad_norms, synth, small_weights
from oracle import mmformer_ref, train_step_ref
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
g = gold("masks15_train.npz"); S = int(g["S"]); dev = torch.device("cuda:0")
x, target = synth(1, S, blobs=True)
keys = ["multimodal_decode_conv", "t1_pos", "t1_encode_conv", "decoder_fuse", "t1_encoder", "multimodal_transformer"]
for row in [int(a) for a in sys.argv[1:]] or [1, 0]:
    ref = dict(zip([str(k) for k in g["mmformer_step_grad_keys"]], g["mmformer_step_grad_norms"][row]))
    mask = torch.from_numpy(g["masks"][row:row + 1]).to(dev)
    for dt in (torch.float32, torch.float64):
        P = {k: v.to(dev, dt).clone().requires_grad_(True) for k, v in small_weights().items()}
        outs = mmformer_ref.forward(P, x.to(dev, dt), mask, target.to(dev), 4.0, "idt", True, True)
        loss, parts = train_step_ref.assemble_loss(outs, target.to(dev), mask, "idt", True)
        loss.backward()
        gn = grad_norms((k, v.grad) for k, v in P.items())
        print(f"row {row} oracle on GPU {str(dt)[6:]}: loss {float(loss):.6f}", {k: f"{gn.get(k, 0.0):.6f}" for k in keys})
    print(f"row {row} fixture (reference, CPU f32)      ", {k: f"{ref[k]:.6f}" for k in keys})
