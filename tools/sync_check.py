"""report host<->device synchronisation points inside one training step (torch sync debug mode) + step time breakdown"""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
from passion_amd import train as ptrain
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", False, torch.bfloat16
m.train()
opt = ptrain.make_optimizer(m)
state = ptrain.LossState(dev)
x, target, mask = ptrain.synthetic_batch(1, S, dev)
for _ in range(8):
    ptrain.train_step(m, opt, x, mask, target, 4.0, state, None)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    ptrain.train_step(m, opt, x, mask, target, 4.0, state, None)
torch.cuda.set_sync_debug_mode("default")
print("sync warnings:", len(w))
for i in w[:20]:
    print("  ", str(i.message)[:150], i.filename, i.lineno)
torch.cuda.synchronize()
# phases
for rep in range(3):
    t0 = time.perf_counter()
    m.is_training = True
    outs = m(x, mask, target=target, temp=4.0)
    t1 = time.perf_counter()
    loss = ptrain.assemble_loss(outs, target, mask, "idt", False, state, False, None)
    t2 = time.perf_counter()
    opt.zero_grad()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print(f"host: fwd {1e3*(t1-t0):.2f}  loss {1e3*(t2-t1):.2f}  bwd {1e3*(t3-t2):.2f}  opt {1e3*(t4-t3):.2f}  drain {1e3*(t5-t4):.2f} ms")
# pipelined: no synchronisation between steps
import gc
for label in ("pipelined", "pipelined, gc disabled"):
    if "gc" in label:
        gc.disable()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    delay = float(os.environ.get("HOST_DELAY_MS", "0")) * 1e-3
    for _ in range(30):
        ptrain.train_step(m, opt, x, mask, target, 4.0, state, None)
        if delay:                                   # is the step host-bound?  burn host time and watch the wall clock
            te = time.perf_counter() + delay
            while time.perf_counter() < te:
                pass
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label}: host {1e3*(t1-t0)/30:.2f} ms/step, wall {1e3*(t2-t0)/30:.2f} ms/step")
