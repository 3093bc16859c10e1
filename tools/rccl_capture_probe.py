"""Can this stack (ROCm 7.2 / torch 2.10 / RCCL) record collectives into a HIP graph?  One GPU, process group on the `nccl` backend at
world size 1 (PASSION_DDP_FORCE=1): every case runs in its own process (a crash is an answer) and prints one line.

  micro     all_reduce / all_gather / batch_isend_irecv (peer = self) under torch.cuda.graph, replayed with changed inputs
  depth     depth.GraphedDepthStep at world 1, no process group: recorded step == eager step (loss, parameters), small volume
  depthc    the same with loopback collectives (halo planes, InstanceNorm sums, gradient buckets through RCCL) inside the recording
  ddp       graph.GraphedStep(record_collectives=True): the data-parallel step with GradReducer's overlapped all-reduce recorded
  ddpp      ... with PASSION (the 4-float relative-preference all-reduce inside the loss assembly)

python tools/rccl_capture_probe.py [case ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MICRO = r'''
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
a = torch.arange(8, device=dev, dtype=torch.float32); out = torch.zeros(8, device=dev)
lo = torch.zeros(1024, device=dev, dtype=torch.bfloat16); hi = torch.zeros_like(lo); r_lo = torch.zeros_like(lo); r_hi = torch.zeros_like(lo)
gat = [torch.zeros(8, device=dev)]
what = os.environ.get("PROBE_OPS", "ar,ara,ag,p2p").split(",")
def body():
    t = a * 2
    if "ar" in what:
        dist.all_reduce(t)
    if "ara" in what:
        w = dist.all_reduce(t, async_op=True); w.wait()
    if "ag" in what:
        dist.all_gather(gat, t)
    elif "agt" in what:
        tmp = torch.empty(1, 8, device=dev)
        dist.all_gather_into_tensor(tmp, t)
        gat[0].copy_(tmp[0])
    else:
        gat[0].copy_(t)
    if "p2p" in what:
        reqs = [dist.P2POp(dist.isend, lo, 0), dist.P2POp(dist.irecv, r_lo, 0), dist.P2POp(dist.isend, hi, 0), dist.P2POp(dist.irecv, r_hi, 0)]
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
    else:
        r_lo.copy_(lo); r_hi.copy_(hi)
    out.copy_(t + gat[0])
for _ in range(2):
    body()
torch.cuda.synchronize()
print("micro eager ok", what, flush=True)
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.synchronize()
print("micro eager on side stream ok", flush=True)
with torch.cuda.graph(g, stream=s, capture_error_mode=os.environ.get("PROBE_MODE", "thread_local")):
    body()
print("micro captured", flush=True)
for k in range(3):
    a.fill_(float(k + 1)); lo.fill_(float(k + 1)); hi.fill_(float(-k - 1))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, torch.full((8,), 4.0 * (k + 1), device=dev)), out
    assert torch.equal(r_lo, lo) and torch.equal(r_hi, hi)
print("RESULT micro", what, "recorded and replayed correctly", flush=True)
dist.destroy_process_group()
'''

DEPTH = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from passion_amd import parallel, train as ptrain, depth
from passion_amd.models import mmformer
coll = os.environ.get("PROBE_COLL") == "1"
ctx = parallel.init()
dev = ctx.device
D, H, W = (int(v) for v in os.environ.get("PROBE_DHW", "32,48,48").split(","))
def build():
    torch.manual_seed(1037)
    mmformer.patch_size = (D // 16, H // 16, W // 16)
    m = mmformer.Model(num_cls=4).to(dev)
    mmformer.patch_size = 5
    m.mask_type, m.use_passion, m.compute_dtype = "idt", False, torch.bfloat16
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m
g = torch.Generator().manual_seed(7)
x = torch.randn(1, 4, D, H, W, generator=g).to(dev)
t = torch.nn.functional.one_hot(torch.randint(0, 4, (1, D, H, W), generator=g), 4).permute(0, 4, 1, 2, 3).double().contiguous().to(dev)
mask = torch.ones(1, 4, dtype=torch.bool, device=dev)
res = []
for graphed in (False, False, True):
    m = build()
    red = parallel.GradReducer(m, ctx)
    opt = ptrain.make_optimizer(m, lr=1e-3)
    shard = depth.DepthCtx(rank=0, world=1, loopback=coll)
    if graphed:
        st = depth.GraphedDepthStep(m, opt, shard, D, reducer=red, warm=2)
        step = lambda: st(x, mask, t)
    else:
        step = lambda: depth.sharded_train_step(m, opt, shard, x, mask, t.view(t.shape), D, reducer=red)
    losses = [float(step()) for _ in range(6)]
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tw = time.perf_counter() - t0
    res.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}, 1e3 * th / n, 1e3 * tw / n))
(le0, pe0, _, _), (le, pe, he, we), (lg, pg, hg, wg) = res
print("eager repeats bit for bit:", le0 == le, len([k for k in pe if not torch.equal(pe[k], pe0[k])]))
bad = [k for k in pe if not torch.equal(pe[k], pg[k])]
print("eager", le, "graph", lg)
print(f"RESULT depth{'c' if coll else ''} {D}x{H}x{W}: losses equal {le == lg}, parameters differing {len(bad)}; eager host {he:.1f} wall {we:.1f} ms, "
      f"graph host {hg:.1f} wall {wg:.1f} ms", le[-1], lg[-1])
if dist.is_initialized():
    dist.destroy_process_group()
'''

DDP = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from _util import build_product_model, synth
from passion_amd import parallel, train as ptrain
from passion_amd.graph import GraphedStep
passion = os.environ.get("PROBE_PASSION") == "1"
ctx = parallel.init()
dev = ctx.device
S = 32
x, target = synth(1, S, blobs=True)
mask = torch.tensor([[True, False, True, True]])
x, target, mask = x.to(dev), target.to(dev), mask.to(dev)
res = []
for graphed in (False, True, True, 2):
    torch.manual_seed(5)
    m = build_product_model(S, small=False).to(dev)
    m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, torch.bfloat16
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    red = parallel.GradReducer(m, ctx, overlap=os.environ.get("PROBE_OVERLAP", "1") == "1")
    assert red.active
    opt = ptrain.make_optimizer(m, lr=1e-3)
    state = ptrain.LossState(dev) if passion else None
    if graphed == 2:         # the eager step on a side stream, as GraphedStep's warm-up runs it
        sst = torch.cuda.Stream()
        def step():
            sst.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sst):
                l_ = ptrain.train_step(m, opt, x, mask, target.view(target.shape), 4.0, state, red)
            torch.cuda.current_stream().wait_stream(sst)
            return l_
    elif graphed:
        gs = GraphedStep(m, opt, 4.0, state, warm=2, reducer=red, record_collectives=True)
        assert gs.full and not gs.hybrid
        step = lambda: gs(x, mask, target.view(target.shape))
    else:
        step = lambda: ptrain.train_step(m, opt, x, mask, target.view(target.shape), 4.0, state, red)
    losses = [float(step()) for _ in range(6)]
    torch.cuda.synchronize()
    res.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}))
(le, pe), (lg, pg), (lg2, pg2), (ls, ps) = res
bad = [k for k in pe if not torch.equal(pe[k], pg[k])]
print("eager", le, "graph", lg)
print("graph run twice equal:", lg == lg2, len([k for k in pg if not torch.equal(pg[k], pg2[k])]), "eager on a side stream == eager:", le == ls,
      len([k for k in pe if not torch.equal(pe[k], ps[k])]), ls)
print("differing:", bad[:12])
print(f"RESULT ddp{'p' if passion else ''}: recorded step with its collectives: losses equal {le == lg}, parameters differing {len(bad)}", le[-1], lg[-1])
dist.destroy_process_group()
'''

CASES = {
    "micro": (MICRO, {}),
    "micro_ar": (MICRO, {"PROBE_OPS": "ar"}),
    "micro_ara": (MICRO, {"PROBE_OPS": "ara"}),
    "micro_ag": (MICRO, {"PROBE_OPS": "ag"}),
    "micro_p2p": (MICRO, {"PROBE_OPS": "p2p"}),
    "micro_agt": (MICRO, {"PROBE_OPS": "agt"}),
    "micro_none": (MICRO, {"PROBE_OPS": "none"}),
    "micro_ar_global": (MICRO, {"PROBE_OPS": "ar", "PROBE_MODE": "global"}),
    "micro_ar_relaxed": (MICRO, {"PROBE_OPS": "ar", "PROBE_MODE": "relaxed"}),
    "depth": (DEPTH, {"PASSION_DDP_FORCE": "0"}),
    "depthc": (DEPTH, {"PASSION_DDP_FORCE": "1", "PROBE_COLL": "1"}),
    "depthc_g": (DEPTH, {"PASSION_DDP_FORCE": "1", "PROBE_COLL": "1", "PASSION_DEPTH_HALO": "gather"}),
    "ddp": (DDP, {"PASSION_DDP_FORCE": "1"}),
    "ddp_noverlap": (DDP, {"PASSION_DDP_FORCE": "1", "PROBE_OVERLAP": "0"}),
    "ddpp": (DDP, {"PASSION_DDP_FORCE": "1", "PROBE_PASSION": "1"}),
}


def main():
    import socket
    names = sys.argv[1:] or list(CASES)
    for name in names:
        base = name.split(":")[0]
        script, env_extra = CASES[base]
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("WORLD_SIZE", None)
        env.update(env_extra)
        if ":" in name:
            env["PROBE_DHW"] = name.split(":")[1]
        try:
            p = subprocess.run([sys.executable, "-c", script, ROOT], env=env, capture_output=True, text=True, timeout=900)
            lines = p.stdout.strip().splitlines()
            tail = ([l for l in lines if l.startswith("RESULT")] or lines or ["(no output)"])[-1]
            print(f"[{name}] rc {p.returncode}: {tail}", flush=True)
            logdir = os.path.join(ROOT, "gpurun_out", "r05")
            os.makedirs(logdir, exist_ok=True)
            with open(os.path.join(logdir, "probe_" + name.replace(":", "_") + ".log"), "w") as f:
                f.write(p.stdout + "\n-------- stderr\n" + p.stderr)
            if p.returncode != 0:
                err = [l for l in p.stderr.strip().splitlines() if "frame #" not in l]
                print("    " + "\n    ".join((lines[-3:] + err)[-14:]), flush=True)
        except subprocess.TimeoutExpired:
            print(f"[{name}] TIMEOUT", flush=True)


if __name__ == "__main__":
    main()
