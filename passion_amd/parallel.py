"""One process per GPU over RCCL/xGMI: replaces the reference's single-process nn.DataParallel
(code/train.py:90).  Samples are independent units (InstanceNorm has no batch statistics, losses are
per-sample), so the data path needs NO collective; the only exchange is the gradient all-reduce.

The reference sums per-sample losses over the batch (train.py:229-231), so the equivalent of its
DataParallel reduce-add is a SUM of gradients across ranks.  Gradients are packed into a few large
flat buckets (xGMI is point-to-point, 7 links per GPU: fewer, larger collectives amortise the
per-call latency; mmformer has 142 MB of f32 gradients) and reduced with async all_reduce calls that
are all in flight together.
"""
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


def _raw_stream():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


@dataclass
class Ctx:
    rank: int
    world: int
    local_rank: int
    device: torch.device
    backend: str


# PASSION_DDP_FORCE=1: the N > 1 code path -- process group, overlap hooks, bucket packing across streams, async all_reduce on
# RCCL's stream, reduced gradients as views of the buckets -- also in a single process (a SUM over one rank is the identity).
# One-GPU boxes can then drive librccl with exactly the calls the 8-GPU run issues (tests/test_rccl_world1_gpu.py,
# bench.py --ddp-mode).
FORCE = os.environ.get("PASSION_DDP_FORCE") == "1"


def init(backend=None):
    """reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun); single process if absent"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        # PASSION_DEVICE_INDEX: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (every rank on that
        # device; RCCL refuses two ranks per GPU, so combine with PASSION_DIST_BACKEND=gloo)
        local_dev = int(os.environ.get("PASSION_DEVICE_INDEX", local))
        torch.cuda.set_device(local_dev)
        device = torch.device("cuda", local_dev)
    else:
        device = torch.device("cpu")
    backend = backend or os.environ.get("PASSION_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    if (world > 1 or FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if use_gpu and backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return Ctx(rank, world, local, device, backend)


WIRE_BF16 = os.environ.get("PASSION_DDP_BF16") == "1"    # gradients travel as bf16 (half the xGMI bytes); default: f32 on the wire


class GradReducer:
    """Bucketed SUM all-reduce of parameter gradients + initial parameter broadcast from rank 0.

    overlap=True (default): every parameter gets a post-accumulate-grad hook; as soon as the last gradient of a bucket
    has been written during backward the bucket is packed and its all_reduce is issued (async, RCCL's own stream), so
    the collectives of the late layers run under the backward kernels of the early ones.  all_reduce_grads() then only
    launches the buckets that never completed (parameters without gradient in this step), waits and unpacks.
    Buckets are filled in reverse registration order ~ the order in which backward produces gradients.

    Streams: the models run their modality encoders on side streams and autograd replays each backward node on its
    forward's stream, so the gradients of one bucket may be written on several streams while the hook that completes the
    bucket runs on only one of them.  Every hook notes the stream it ran on; the launch makes its own stream wait for an
    event recorded on each of the others (all their gradient kernels are already enqueued by then, hooks fire in host
    order) before it packs the bucket."""

    def __init__(self, model, ctx, bucket_mb=32, overlap=True):
        self.ctx = ctx
        self._names = {id(p): k for k, p in model.named_parameters()}
        self._stream_objs = {}
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_mb * (1 << 20):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.wire_dtype = torch.bfloat16 if (WIRE_BF16 and self.params[0].is_cuda) else torch.float32
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=self.wire_dtype, device=self.params[0].device)
                     for b in self.buckets]
        self.active = ctx.world > 1 or (FORCE and dist.is_initialized())
        self.overlap = overlap and self.active and hasattr(torch.Tensor, "register_post_accumulate_grad_hook")
        self._pending = [len(b) for b in self.buckets]       # gradients still missing per bucket in this backward
        self._works = [None] * len(self.buckets)
        self._next = 0                                       # first bucket that has not been launched in this backward
        self._streams = [set() for _ in self.buckets]        # streams that produced gradients of the bucket in this backward
        self._on_gpu = self.params[0].is_cuda
        self.suspended = False          # hooks off: a recorded forward + backward (graph.GraphedStep, hybrid) -- every bucket is then
                                        # launched by all_reduce_grads(), after the replay
        if self.active:
            for p in model.parameters():
                dist.broadcast(p.data, src=0)
        if self.overlap:
            for bi, bucket in enumerate(self.buckets):
                for p in bucket:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))

    def _make_hook(self, bi):
        def hook(_param):
            if self.suspended:
                return
            if self._on_gpu:
                self._streams[bi].add(_raw_stream())     # raw handle: a Stream object costs 10 us to build, per parameter
            self._pending[bi] -= 1
            # Collectives must be issued in the same order on every rank.  Readiness order can differ between ranks (a
            # parameter that got no gradient on one of them), so buckets are launched strictly by index: bucket k goes out
            # once it is complete AND every bucket before it has gone out; the rest follows in all_reduce_grads().
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)
                self._next += 1
        return hook

    def _stream_of(self, raw):
        """torch Stream for a raw handle noted by a hook.  The default stream reports handle 0, but an event recorded on
        torch.cuda.ExternalStream(0) does NOT wait for the default stream's pending work on this stack (152 of 200 trials in
        tools/null_stream_event.py ran ahead): that was the stale-bucket race behind the flaky 2-rank test.  Handle 0 maps back to
        torch.cuda.default_stream(); the side streams' handles are real pointers and wrap fine."""
        st = self._stream_objs.get(raw)
        if st is None:
            d = torch.cuda.default_stream(self.params[0].device)
            st = d if raw == d.cuda_stream else torch.cuda.ExternalStream(raw)
            self._stream_objs[raw] = st
        return st

    def _launch(self, bi):
        bucket, flat = self.buckets[bi], self.flat[bi]
        if self._on_gpu:
            cur = torch.cuda.current_stream()
            for raw in self._streams[bi]:
                if raw != cur.cuda_stream:
                    ev = torch.cuda.Event()
                    ev.record(self._stream_of(raw))
                    cur.wait_event(ev)
            self._streams[bi].clear()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
        torch._foreach_copy_(list(flat.split([g.numel() for g in grads])), [g.reshape(-1) for g in grads])
        self._works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def all_reduce_grads(self):
        if not self.active:
            return
        for bi in range(self._next, len(self.buckets)):  # not (completely) produced by this backward, or overlap off
            self._launch(bi)
        for bi, (bucket, flat) in enumerate(zip(self.buckets, self.flat)):
            self._works[bi].wait()
            views = flat.split([p.numel() for p in bucket])
            if self.wire_dtype != torch.float32:             # bf16 on the wire: back to f32 gradients, one multi-tensor launch
                have = [(p, v) for p, v in zip(bucket, views) if p.grad is not None]
                torch._foreach_copy_([p.grad.view(-1) for p, _ in have], [v for _, v in have])
                for p, v in zip(bucket, views):
                    if p.grad is None:
                        p.grad = v.view_as(p).float()
                continue
            # f32: the reduced gradient IS the bucket slice -- p.grad becomes a view of the flat buffer (no copy back; the
            # optimizer reads it, zero_grad(set_to_none) drops it, the next backward's pack overwrites the buffer after that)
            for p, v in zip(bucket, views):
                p.grad = v.view_as(p)
        self._pending = [len(b) for b in self.buckets]
        self._works = [None] * len(self.buckets)
        self._next = 0


def all_reduce_sum_(t):
    """in-place SUM of a (small, detached) tensor over all ranks; a no-op in a single process.  Used for the two places where
    the reference's loss assembly couples the samples of the GLOBAL batch (DataParallel gathers the per-sample outputs before
    train.py:266-270 forms rp_iter, and :325 sums epoch_dist_m over every sample of the epoch)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def barrier(ctx):
    if ctx.world > 1:
        dist.barrier()


def all_reduce_max(ctx, value):
    if ctx.world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=ctx.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
