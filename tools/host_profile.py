"""where does the host spend the step?  cProfile over K training steps (plain or PASSION) at 128^3 -> top functions by own time"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import train as ptrain, optim
from passion_amd.models import mmformer
S = int(os.environ.get("S", "128"))
passion = len(sys.argv) > 1 and sys.argv[1] == "passion"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
torch.manual_seed(0)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype, m.is_training = "idt", passion, torch.bfloat16, True
m.train()
opt = optim.FusedAdamW(m.parameters(), lr=2e-4, weight_decay=1e-4, amsgrad=True) if hasattr(optim, "FusedAdamW") else torch.optim.AdamW(m.parameters(), lr=2e-4)
x, target, _ = ptrain.synthetic_batch(1, S, dev)
mask = torch.tensor([[True, True, True, True]], device=dev)


def step():
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", passion)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
if os.environ.get("PASSION_NOGC") == "1":
    import gc
    gc.collect(); gc.freeze(); gc.disable()
import time
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, wall {1e3 * (t2 - t0) / K:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step()
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
ps = pstats.Stats(pr, stream=st)
ps.sort_stats("tottime").print_stats(45)
out = st.getvalue()
print("\n".join(l[:170] for l in out.splitlines()))
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(60)
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"host_profile_{'passion' if passion else 'plain'}_cum.txt"), "w").write(st.getvalue())
