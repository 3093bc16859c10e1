import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
from passion_amd import train as ptrain, ops

S = 64
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", False, torch.bfloat16
m.eval()   # no dropout: replays must be bit-identical
opt = ptrain.make_optimizer(m)
x, target, mask = ptrain.synthetic_batch(1, S, dev)
m.is_training = True

def fwd_bwd():
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", False)
    loss.backward()
    return loss

s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        opt.zero_grad(set_to_none=False); l = fwd_bwd(); opt.step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
opt.zero_grad(set_to_none=False); le = fwd_bwd(); torch.cuda.synchronize()
ge = {k: p.grad.clone() for k, p in m.named_parameters()}
print("eager loss", float(le))
g = torch.cuda.CUDAGraph(); opt.zero_grad(set_to_none=False)
with torch.cuda.graph(g):
    sl = fwd_bwd()
for r in range(3):
    opt.zero_grad(set_to_none=False); g.replay(); torch.cuda.synchronize()
    bad = [(k, float((p.grad - ge[k]).abs().max()), float(ge[k].abs().max())) for k, p in m.named_parameters() if not torch.allclose(p.grad, ge[k], rtol=1e-2, atol=1e-4)]
    print(f"replay {r}: loss {float(sl):.5f}; params with different grads: {len(bad)}", bad[:4])
opt.step(); torch.cuda.synchronize()
opt.zero_grad(set_to_none=False); le2 = fwd_bwd(); torch.cuda.synchronize(); print("eager after step", float(le2))
opt.zero_grad(set_to_none=False); g.replay(); torch.cuda.synchronize(); print("replay after step", float(sl))
