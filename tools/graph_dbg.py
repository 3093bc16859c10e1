import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
from passion_amd import train as ptrain, graph as pgraph, parallel
variant = sys.argv[1]
S = 128
if "init" in variant:
    ctx = parallel.init(); dev = ctx.device
else:
    dev = torch.device("cuda:0"); ctx = parallel.Ctx(0, 1, 0, dev, "nccl")
mmformer.patch_size = S // 16
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype = "idt", False, torch.bfloat16
m.train()
if "reducer" in variant:
    red = parallel.GradReducer(m, ctx)
opt = ptrain.make_optimizer(m, capturable=True)
state = ptrain.LossState(dev)
x, target, mask = ptrain.synthetic_batch(1, S, dev, seed=7)
gs = pgraph.GraphedTrainStep(m, opt, x, mask, target, temp=4.0, state=state)
out = []
step = (lambda: gs(x, mask, target)) if "lambda" in variant else None
for i in range(0 if 'check' in variant else 6):
    loss = step() if step else gs(x, mask, target)
    if "skipf" in variant and i == 1:
        continue
    if "floatall" in variant and i == 1:
        out.append(f"{float(loss):.4f}")
        torch.cuda.synchronize()
        continue
    if "streamsync" in variant and i == 1:
        torch.cuda.current_stream().synchronize()
        continue
    if "devsync" in variant and i == 1:
        torch.cuda.synchronize()
        if "gc" in variant:
            import gc; gc.collect()
        if "perf" in variant:
            parallel.barrier(ctx); t0 = time.perf_counter()
        continue
    if "nosync" not in variant or i >= 1:
        out.append(f"{float(loss):.4f}")
print(variant, " ".join(out))
if "check" in variant:
    import itertools
    for it in range(12):
        loss = gs(x, mask, target)
        if it % 2 == 1:
            torch.cuda.current_stream().synchronize()
        bad = []
        gmax = 0.0
        for name, p_ in m.named_parameters():
            if p_.grad is None:
                continue
            g = p_.grad.float()
            v = float(g.abs().max())
            if not (v == v) or v > 1e4:
                bad.append((name, v))
        print(f"replay {it}: loss {float(loss):.4f} bad grads: {len(bad)} {bad[:6]}", flush=True)
        if bad:
            break
