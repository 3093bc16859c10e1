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
gs = pgraph.GraphedTrainStep(m, opt, x, mask, target, temp=4.0, state=state, prof_mask=int(os.environ.get('PROF', '0')))
print("captured", flush=True)
for i in range(int(os.environ.get('STEPS', '8'))):
    torch.cuda.synchronize(); t0 = time.time()
    loss = gs(x, mask, target)
    torch.cuda.synchronize(); t1 = time.time()
    print(f"graph step {1e3*(t1-t0):.2f} ms loss {float(loss):.4f}", flush=True)
from passion_amd import ops
recs = ops.prof_read(keep=True)
print("graph-recorded prof records:", len(recs), "timed:", sum(1 for r in recs if r[1] >= 0))
names = {0: "fwd_k3", 1: "fwd_k1", 2: "dgrad", 3: "wgrad"}
for kind in range(4):
    r = [x for x in recs if x[0] == kind and x[1] >= 0]
    if r:
        ms = sum(x[1] for x in r)
        print(f"{names[kind]}: n={len(r)} total {ms:.2f} ms  {sum(x[2] for x in r)/ms/1e6:.1f} GB/s")
