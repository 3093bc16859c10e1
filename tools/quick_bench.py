"""ad-hoc: time fwd/bwd of the product model at a given size (not the official bench)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
from passion_amd import train as ptrain, ops

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
passion = len(sys.argv) > 3 and sys.argv[3] == "passion"
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, dtype
m.train()
opt = ptrain.make_optimizer(m)
x, target, mask = ptrain.synthetic_batch(1, S, dev)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.time()
    m.is_training = True
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", passion)
    t1h = time.time()
    torch.cuda.synchronize(); t1 = time.time()
    opt.zero_grad(); loss.backward()
    t2h = time.time()
    torch.cuda.synchronize(); t2 = time.time()
    opt.step(); torch.cuda.synchronize(); t3 = time.time()
    print(f"S={S} {dtype} passion={passion} it{i}: fwd {1e3*(t1-t0):.1f} ms bwd {1e3*(t2-t1):.1f} ms opt {1e3*(t3-t2):.1f} ms [host enqueue fwd {1e3*(t1h-t0):.1f} bwd {1e3*(t2h-t1):.1f}] loss {float(loss):.4f} mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
ops.prof_enable(True)
m.is_training = True
outs = m(x, mask, target=target, temp=4.0)
loss = ptrain.assemble_loss(outs, target, mask, "idt", passion); loss.backward(); torch.cuda.synchronize()
ops.prof_enable(False)
recs = ops.prof_read()
names = {0: "fwd_k3", 1: "fwd_k1", 2: "dgrad", 3: "wgrad"}
for k in range(4):
    r = [x for x in recs if x[0] == k]
    if r:
        ms = sum(x[1] for x in r); by = sum(x[2] for x in r); fl = sum(x[3] for x in r)
        print(f"{names[k]}: n={len(r)} total {ms:.2f} ms  {by/ms/1e6:.1f} GB/s  {fl/ms/1e9:.2f} TFLOP/s")
