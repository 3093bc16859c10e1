"""every timed conv / finish / attention launch of ONE plain training step (mmformer 4 x S^3, bf16) on a single stream, longest first:
algorithmic MB, us, GB/s, fraction of 8 TB/s.  usage: launch_table.py [S] [top] [passion]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd.models import mmformer
from passion_amd import train as ptrain, ops

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
top = int(sys.argv[2]) if len(sys.argv) > 2 else 70
passion = len(sys.argv) > 3 and sys.argv[3] == "passion"
dev = torch.device("cuda:0")
mmformer.patch_size = S // 16
mmformer.ENC_STREAMS = False
torch.manual_seed(1037)
m = mmformer.Model().to(dev)
m.mask_type, m.use_passion, m.compute_dtype, m.is_training = "idt", passion, torch.bfloat16, True
m.train()
opt = ptrain.make_optimizer(m)
x, target, mask = ptrain.synthetic_batch(1, S, dev)


def step():
    outs = m(x, mask, target=target, temp=4.0)
    loss = ptrain.assemble_loss(outs, target, mask, "idt", passion)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
ops.prof_read()
ops.prof_enable(0x7F)
step()
torch.cuda.synchronize()
ops.prof_enable(0)
recs = [r for r in ops.prof_read() if r[1] >= 0]
kinds = {0: "fwd3", 1: "fwd1", 2: "dgrad", 3: "wgrad", 4: "attn", 5: "attn", 6: "finish"}
tot = sum(r[1] for r in recs)
print(f"{len(recs)} timed launches, {tot:.2f} ms on one stream")
small = [r for r in recs if r[2] < 40e6]
print(f"launches that move < 40 MB: {len(small)} launches, {sum(r[1] for r in small):.2f} ms ({sum(r[2] for r in small) / 1e6:.0f} MB algorithmic)")
fam = {}
for r in recs:
    f = fam.setdefault((kinds.get(r[0], str(r[0])), r[4]), [0, 0.0, 0.0])
    f[0] += 1; f[1] += r[1]; f[2] += r[2]
print("-- by kernel")
for (k, n), (c, ms, by) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{ms * 1e3:8.1f} us {c:3d}x {k:6s} {by / 1e6:8.1f} MB {by / ms / 1e6 if ms > 0 else 0:7.0f} GB/s {by / ms / 8e9 * 100 if ms > 0 else 0:5.1f}%  {n[:70]}")
print("-- by launch")
for r in sorted(recs, key=lambda r: -r[1])[:top]:
    print(f"{r[1] * 1e3:8.1f} us {kinds.get(r[0], str(r[0])):6s} {r[2] / 1e6:8.1f} MB {r[2] / r[1] / 1e6:7.0f} GB/s {r[2] / r[1] / 8e9 * 100:5.1f}%  {r[4][:70]}")
