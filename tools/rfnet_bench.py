"""RFNet(+PASSION) step timing: python tools/rfnet_bench.py S B [passion]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import rfnet
from passion_amd import train as ptrain, ops
S, B = int(sys.argv[1]), int(sys.argv[2]); passion = "passion" in sys.argv
dev = torch.device("cuda:0"); torch.manual_seed(1037)
m = rfnet.Model().to(dev); m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, torch.bfloat16; m.train()
opt = ptrain.make_optimizer(m); state = ptrain.LossState(dev)
x, target, mask = ptrain.synthetic_batch(B, S, dev)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    m.is_training = True
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", passion, state)
    torch.cuda.synchronize(); t1 = time.time()
    opt.zero_grad(); loss.backward(); torch.cuda.synchronize(); t2 = time.time()
    opt.step(); torch.cuda.synchronize(); t3 = time.time()
    print(f"rfnet S={S} B={B} passion={passion} it{i}: fwd {1e3*(t1-t0):.1f} bwd {1e3*(t2-t1):.1f} opt {1e3*(t3-t2):.1f} ms loss {float(loss):.4f}", flush=True)
