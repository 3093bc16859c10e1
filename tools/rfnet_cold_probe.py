"""tests/test_fullsize_gpu.py::test_rfnet_128_batch2_passion_training_step fails when it is the FIRST thing a process does (5 of 6 runs,
round-5 tree included) and passes inside the full suite: which gradients differ between the first (cold) step of a process and the
following ones, and by how much?  python tools/rfnet_cold_probe.py [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import train as ptrain
from passion_amd.models import rfnet
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
torch.manual_seed(1037)
m = rfnet.Model(num_cls=4).to(dev)
m.mask_type, m.compute_dtype, m.use_passion, m.is_training = "idt", torch.bfloat16, True, True
m.eval()
x, t, _ = ptrain.synthetic_batch(2, S, dev)
mask = torch.tensor([[True, True, False, True], [False, True, True, True]], device=dev)
from passion_amd import ops
STK = []
_unstack = ops.unstack
def _hooked(t, n):
    if t.requires_grad:
        lvl = len(STK_CUR)
        t.register_hook(lambda g, lvl=lvl: (torch.cuda.synchronize() if os.environ.get("PROBE_SYNC_STK") else None, STK_CUR.append((lvl, g.detach().clone())))[1])
        STK_CUR.append(None)
    return _unstack(t, n)
ops.unstack = _hooked
import passion_amd.models.rfnet as _rf
STK_CUR = []
def step():
    global STK_CUR
    STK_CUR = []
    r = _step()
    STK.append([e for e in STK_CUR if e is not None])
    return r
def _join():
    pr = os.environ.get("PROBE_PAIR")
    if pr:
        side = m.__dict__.get("_streams_dev", {}).get(dev, [])
        nm = {"main": torch.cuda.current_stream(dev), "s0": side[0], "s1": side[1], "s2": side[2], "s3": side[3]}
        for item in pr.split(","):
            a, b2 = item.split(">")
            nm[a].wait_stream(nm[b2])
        return
    if os.environ.get("PROBE_JOIN") == "gpu":          # all-to-all stream waits on the GPU instead of a host sync
        ss = list({id(s_): s_ for s_ in m.__dict__.get("_streams_dev", {}).get(dev, [])}.values()) + [torch.cuda.current_stream(dev), torch.cuda.default_stream(dev)]
        for a in ss:
            for b2 in ss:
                if a is not b2:
                    a.wait_stream(b2)
    else:
        torch.cuda.synchronize()
def _step():
    m.zero_grad(set_to_none=True)
    outs = m(x, mask, target=t, temp=4.0)
    st = ptrain.LossState(dev, modal_weight=torch.tensor([2.43, 1.62, 1.19, 5.09]))
    loss = ptrain.assemble_loss(outs, t, mask, "idt", True, st)
    if os.environ.get("PROBE_SYNC_FWD"):
        torch.cuda.synchronize()
    if os.environ.get("PROBE_SYNC_OUTS"):          # a device sync whenever a gradient reaches one of the model's outputs
        which = os.environ.get("PROBE_SYNC_OUTS")
        for oi, o in enumerate(outs):
            if o.requires_grad and (which == "1" or str(oi) in which.split(",")):
                o.register_hook(lambda g: (_join(), g)[1])
    loss.backward()
    torch.cuda.synchronize()
    return [o.detach().clone() for o in outs], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
runs = [step() for _ in range(4)]
for i in range(1, 4):
    o_eq = all(torch.equal(a, b) for a, b in zip(runs[0][0], runs[i][0]))
    diff = [(k, float((runs[0][1][k] - runs[i][1][k]).abs().max()), float(runs[0][1][k].abs().max())) for k in runs[0][1] if not torch.equal(runs[0][1][k], runs[i][1][k])]
    print(f"run 0 vs run {i}: outputs equal {o_eq}; {len(diff)} of {len(runs[0][1])} gradients differ")
    for k, d, a in diff[:12]:
        print(f"    {k}: max |diff| {d:.3e} of max |g| {a:.3e}")
d12 = [k for k in runs[1][1] if not torch.equal(runs[1][1][k], runs[2][1][k])]
print(f"run 1 vs run 2: {len(d12)} gradients differ")

for i in range(1, 4):
    for (l0, g0), (l1, g1) in zip(sorted(STK[0], key=lambda e: e[0]), sorted(STK[i], key=lambda e: e[0])):
        eq = torch.equal(g0, g1)
        msg = ""
        if not eq:
            dm = g0 != g1
            a, b2 = g0[dm].float(), g1[dm].float()
            rel = ((a - b2).abs() / (a.abs() + b2.abs() + 1e-30))
            idx = dm.reshape(dm.shape[0], -1).nonzero()
            msg = (f" n diff {int(dm.sum())} of {g0.numel()}; |g| at those: median {float(a.abs().median()):.3e} (tensor max {float(g0.float().abs().max()):.3e}); "
                   f"relative diff median {float(rel.median()):.3e} max {float(rel.max()):.3e}; samples hit {sorted(set(idx[:, 0].tolist()))}; "
                   f"flat index range {int(idx[:, 1].min())}..{int(idx[:, 1].max())}")
        print(f"stk grad marker {l0}: run 0 vs {i} equal {eq}" + msg)
