"""whole-step HIP graph vs eager: python tools/graph_step_bench.py S [passion] [eval]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
from passion_amd import train as ptrain, graph as pgraph

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
passion = "passion" in sys.argv
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, torch.bfloat16
m.train()
if "eval" in sys.argv:
    m.eval()
opt = ptrain.make_optimizer(m, capturable=True)
state = ptrain.LossState(dev)
x, target, mask = ptrain.synthetic_batch(1, S, dev)
gs = pgraph.GraphedTrainStep(m, opt, x, mask, target, temp=4.0, state=state)
print("captured", flush=True)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.time()
    loss = gs(x, mask, target)
    torch.cuda.synchronize(); t1 = time.time()
    print(f"graph step {1e3*(t1-t0):.2f} ms loss {float(loss):.4f}", flush=True)
