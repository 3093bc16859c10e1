"""experiment: capture forward + loss + backward of the product model in a HIP graph (torch.cuda.graph) and replay"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
from passion_amd import train as ptrain, ops

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
passion = len(sys.argv) > 2 and sys.argv[2] == "passion"
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, torch.bfloat16
m.train()
opt = ptrain.make_optimizer(m)
x, target, mask = ptrain.synthetic_batch(1, S, dev)
m.is_training = True

def fwd_bwd():
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", passion)
    loss.backward()
    return loss

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=False)
        fwd_bwd()
        opt.step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=False)
with torch.cuda.graph(g):
    static_loss = fwd_bwd()
torch.cuda.synchronize()
print("captured")
for i in range(6):
    torch.cuda.synchronize(); t0 = time.time()
    for p in m.parameters():
        pass
    opt.zero_grad(set_to_none=False)
    g.replay()
    torch.cuda.synchronize(); t1 = time.time()
    opt.step()
    torch.cuda.synchronize(); t2 = time.time()
    print(f"graph replay fwd+bwd {1e3*(t1-t0):.1f} ms opt {1e3*(t2-t1):.1f} ms loss {float(static_loss):.4f}", flush=True)
