"""librccl driven by this code on a one-GPU box: the `nccl` backend initialised at world size 1 (PASSION_DDP_FORCE=1) runs exactly
the calls every rank of an 8-GPU job issues -- init_process_group(device_id), parameter broadcast, GradReducer's overlap hooks with
gradients arriving from the model's side streams, cross-stream event waits, bucket packing, all_reduce(async_op=True) on RCCL's stream,
reduced gradients as views of the buckets, the 4-float relative-preference all-reduce; and depth.DepthCtx.exchange's grouped
send / recv pairs (batch_isend_irecv) on device buffers, peer = this rank.  A SUM over one rank is the identity, so the step must equal
the single-process step BIT FOR BIT (bf16: every sum on that path is order-independent)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from _util import build_product_model, synth
from passion_amd import parallel, train as ptrain, depth

force = os.environ.get("PASSION_DDP_FORCE") == "1"
ctx = parallel.init()
assert ctx.backend == "nccl" and dist.is_initialized() == force
dev = ctx.device
S = 32
m = build_product_model(S, small=False).to(dev)      # full channel counts: every sum on the bf16 path is order-independent there
m.mask_type, m.use_passion, m.compute_dtype = "idt", True, torch.bfloat16
m.train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
red = parallel.GradReducer(m, ctx)
assert red.active == force and red.overlap == force
opt = ptrain.make_optimizer(m, lr=1e-3)
state = ptrain.LossState(dev)
x, target = synth(2, S, blobs=True)
mask = torch.tensor([[True, False, True, True], [True, True, True, False]])
x, target, mask = x.to(dev), target.to(dev), mask.to(dev)
losses = []
if os.environ.get("W1_GRAPH") == "1":
    # the N > 1 form of graph.GraphedStep: forward + loss + backward replayed, RCCL all-reduce + AdamW eager after every replay
    from passion_amd.graph import GraphedStep
    m.use_passion = False
    gs = GraphedStep(m, opt, 4.0, None, warm=1, reducer=red)
    assert gs.hybrid == force
    for it in range(4):
        losses.append(float(gs(x, mask, target.view(target.shape))))
    assert gs.replays == 3
else:
    if os.environ.get("W1_PLAIN") == "1":
        m.use_passion = False
    for it in range(4 if os.environ.get("W1_PLAIN") == "1" else 3):
        losses.append(float(ptrain.train_step(m, opt, x, mask, target.view(target.shape), 4.0, state if m.use_passion else None, red)))
torch.cuda.synchronize()
if force:
    # gradients of the last step are views of the reducer's flat buckets (no copy back)
    flat_ptrs = [(f.data_ptr(), f.data_ptr() + f.numel() * f.element_size()) for f in red.flat]
    assert all(any(lo <= p.grad.data_ptr() < hi for lo, hi in flat_ptrs) for p in m.parameters() if p.grad is not None)
    # halo planes through batch_isend_irecv on RCCL, peer = self
    sh = depth.DepthCtx(rank=0, world=1, loopback=True)
    a = torch.randn(2, 4096, device=dev).to(torch.bfloat16); b = torch.randn(2, 4096, device=dev).to(torch.bfloat16)
    assert sh.exchange(a, b) == (None, None)
    torch.cuda.synchronize()
    r_lo, r_hi = sh.last_loopback
    assert torch.equal(r_lo, a) and torch.equal(r_hi, b)
    t = torch.arange(4, device=dev, dtype=torch.float32)
    assert torch.equal(parallel.all_reduce_sum_(t.clone()), t)
    dist.barrier()
    dist.destroy_process_group()
sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
torch.save({"losses": losses, "sd": sd, "beta": state.epoch_dist_m.cpu()}, sys.argv[2])
print("done", losses)
'''


def _run(tmp, force, extra=None):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env["PASSION_DDP_FORCE"] = "1" if force else "0"
    env.update(extra or {})
    out = os.path.join(tmp, f"r{int(force)}{len(extra or {})}{'g' if (extra or {}).get('W1_GRAPH') else ''}.pt")
    p = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, out], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    return torch.load(out, weights_only=True)


def test_ddp_code_path_on_rccl_at_world_size_one_equals_the_single_process_step(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, b = _run(str(tmp_path), True), _run(str(tmp_path), False)
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    bad = [k for k in a["sd"] if not torch.equal(a["sd"][k], b["sd"][k])]
    assert not bad, bad[:5]
    assert torch.equal(a["beta"], b["beta"])


def test_hybrid_graph_step_with_eager_rccl_tail_equals_the_eager_step(tmp_path):
    """what bench.py runs for N > 1: GraphedStep(reducer=...) -- the recording holds forward + loss + backward, every replay is followed
    by the bucketed all-reduce (RCCL, world size 1) and AdamW, eagerly -- against four eager single-process steps, bit for bit"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a = _run(str(tmp_path), True, {"W1_GRAPH": "1"})
    b = _run(str(tmp_path), False, {"W1_PLAIN": "1"})
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    bad = [k for k in a["sd"] if not torch.equal(a["sd"][k], b["sd"][k])]
    assert not bad, bad[:5]
