"""One training iteration recorded as a HIP graph and replayed.

The reference's hot loop (code/train.py:222-289: forward, loss assembly, backward, optimizer.step()) issues the same launches in
the same order every iteration; on the MI355X build that is ~620 launches on five streams, and the host needs ~11 ms to enqueue
what the GPU executes in ~12.  `GraphedStep` keeps train.train_step as the ONLY definition of the step -- it calls it
eagerly for the first `warm` iterations (allocator pools, per-stream scratch of the library, LDS grants, packed weight images), then
records one call of it with torch.cuda.graph (stream capture: the side streams of the model become branches of the graph) and
replays the recording from then on.  What differs between iterations lives in device memory:
  * the batch: x / mask / target are copied into the static tensors the recording reads (the label volume is converted to its f32
    channels-last image inside the recording, every replay);
  * AdamW's step-dependent scalars and the learning rate: passion_adamw_step_dev reads them from device words that
    FusedAdamW.graph_tick() rewrites before every replay (optim.py), so `param_groups[i]["lr"]` may change between replays;
  * attention dropout: the kernels add a 64-bit device word to the seed they were recorded with (passion_set_step_words); the word is
    redrawn from torch's CPU generator before every replay.  torch's own dropout (token dropout of the transformers) is handled by
    torch's graph-safe Philox state.
Everything else (modality masks, imb_beta / modal_weight, the relative-preference gate, the epoch accumulators) already was
device-side data (train.assemble_loss has no host synchronisation).

More than one rank (`reducer` given and active): HYBRID -- the recording holds forward + loss + backward only (no collective is
recorded: the reducer's hooks are suspended), every replay is followed by the eager tail of train.train_step: the bucketed RCCL
all-reduce of the (static) gradient tensors and the optimizer step.  The compute keeps the graph's launch cost (the eager N > 1 step
is host-bound: 14.6 ms of enqueue on a slow host against 11 ms of GPU work), the collectives are the same eager calls the tests cover.
The all-reduce is not overlapped with the backward in this mode (PASSION_DDP_BF16=1 halves its bytes).  Plain training only: PASSION's
relative-preference gate is a 4-float all-reduce inside the loss assembly, such steps stay eager for N > 1.
"""
import ctypes
import os
from ctypes import c_void_p

import torch

from . import _lib as L
from . import train as ptrain


class GraphedStep:
    def __init__(self, model, optimizer, temp=4.0, state=None, warmup=False, modal_num=None, iter_per_epoch=None, warm=3, reducer=None,
                 step_fn=None, record_collectives=None):
        if not hasattr(optimizer, "begin_capture"):
            raise L.PassionHipError("GraphedStep needs passion_amd.optim.FusedAdamW (train.make_optimizer on a GPU model)")
        self.model, self.optimizer = model, optimizer
        active = reducer is not None and getattr(reducer, "active", False)
        # record_collectives: the RCCL calls of the step (gradient buckets, PASSION's 4-float gate, the depth path's halo planes and
        # InstanceNorm sums) are recorded INTO the graph -- torch's ProcessGroupNCCL issues them on its own stream behind events, which
        # stream capture follows like any other fork (tools/rccl_capture_probe.py: all_reduce / all_gather / batch_isend_irecv under
        # torch.cuda.graph on this stack).  Default (None): PASSION_GRAPH_COLLECTIVES=1 or a custom step body.
        if record_collectives is None:
            record_collectives = step_fn is not None or os.environ.get("PASSION_GRAPH_COLLECTIVES") == "1"
        self.full = bool(active and record_collectives)          # N > 1, everything in ONE recording
        self.hybrid = active and not self.full
        if self.hybrid and model.use_passion:
            raise L.PassionHipError("GraphedStep (hybrid) with a gradient reducer records plain training steps only: PASSION's gate is a "
                                    "collective (record_collectives=True / PASSION_GRAPH_COLLECTIVES=1 records it)")
        self.reducer = reducer
        self.step_fn = step_fn           # callable(x, mask, target) -> loss: ONE whole iteration (depth.GraphedDepthStep); None = train.train_step
        self.state = state
        self.kw = dict(temp=temp, state=state, reducer=reducer if active else None, warmup=warmup, modal_num=modal_num,
                       iter_per_epoch=iter_per_epoch)
        # the first calls run eagerly on the recording's stream: AdamW creates its moment tensors, the library its per-stream scratch, the
        # conv weights their packed images.  Recorded instead, those allocations and their zero-fills would become graph nodes (every
        # replay re-zeroing the moments) -- so at least one eager call is not negotiable
        self.warm = max(1, int(warm))
        self.calls = 0
        self.graph = None
        self.stream = None
        self.sig = None
        self.replays = 0

    # ------------------------------------------------------------------------------------------
    def _signature(self, x, mask, target):
        m = self.model
        st = self.state
        # the recording holds ADDRESSES: of the loss state's tensors (LossState updates them in place; a rebound tensor would leave the
        # replay reading a freed allocation) -- and values that were python scalars when it was made (warmup, temp)
        st_sig = None if st is None else (st.imb_beta.data_ptr(), st.modal_weight.data_ptr(), st.epoch_dist_m.data_ptr())
        return (tuple(x.shape), x.dtype, tuple(mask.shape), tuple(target.shape), target.dtype, m.mask_type, bool(m.use_passion),
                m.compute_dtype, bool(m.training), bool(self.kw["warmup"]), float(self.kw["temp"]), st_sig)

    def _eager(self, x, mask, target):
        dev = x.device
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            loss = self.step_fn(x, mask, target) if self.step_fn is not None else \
                ptrain.train_step(self.model, self.optimizer, x, mask, target, **self.kw)
        cur.wait_stream(self.stream)
        return loss

    def _record(self, x, mask, target):
        dev = x.device
        lib = L.load()
        self.sx, self.smask, self.starget = x.clone(), mask.to(device=dev, dtype=torch.bool).clone(), target.clone()
        self.words = torch.zeros(32, dtype=torch.float32, device=dev)      # [0, 2): dropout seed offset; [16, 32): AdamW scalars
        self.hyper = self.words[16:]
        torch.cuda.synchronize(dev)
        dot = os.environ.get("PASSION_GRAPH_DOT")          # developer switch: the recording as a DOT file (nodes, edges: what depends on what)
        g = torch.cuda.CUDAGraph(keep_graph=True) if dot else torch.cuda.CUDAGraph()
        lib.passion_set_step_words(c_void_p(self.words.data_ptr()))
        if len(self.optimizer.state) == 0:
            raise L.PassionHipError("GraphedStep: the optimizer has no state yet -- its moment tensors would be created (and zero-filled on "
                                    "every replay) inside the recording; run an eager step first")
        if not self.hybrid:
            self.optimizer.begin_capture(self.hyper)
        else:
            self.reducer.suspended = True
        import torch.distributed as dist
        pg_up = dist.is_available() and dist.is_initialized()
        if pg_up and dist.get_backend() == "nccl":
            # the collective backend's watchdog thread polls the end events of the eager steps' collectives (every 100 ms); an event query
            # that meets a stream which has meanwhile begun capturing is refused on this stack (hipErrorCapturedEvent) and takes the
            # process down (tools/rccl_capture_probe.py depthc at 32 x 48 x 48: the small volume's warm-up ends milliseconds before the
            # recording starts).  Everything is complete (synchronize above): let the watchdog retire it first
            import time
            time.sleep(0.35)
        ok = False
        try:
            # a process group is up: thread-local capture rules, so that what the collective backend's watchdog thread
            # calls meanwhile (event queries on finished work) is neither refused nor able to invalidate this recording
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local" if pg_up else "global"):
                # a view: ops.target_to_cl keeps the converted copy on the tensor OBJECT it is given
                tv = self.starget.view(self.starget.shape)
                if self.step_fn is not None:
                    self.loss = self.step_fn(self.sx, self.smask, tv)
                else:
                    self.loss = ptrain.train_step(self.model, self.optimizer, self.sx, self.smask, tv,
                                                  phase="fb" if self.hybrid else "all", **self.kw)
            ok = True
        finally:
            lib.passion_set_step_words(None)
            if not self.hybrid:
                self.cap = self.optimizer.end_capture()
            elif not ok:
                self.reducer.suspended = False       # no recording: the caller's eager steps keep their overlapped all-reduce
        if dot and ok:
            hip = ctypes.CDLL("libamdhip64.so")
            rc = hip.hipGraphDebugDotPrint(c_void_p(g.raw_cuda_graph()), dot.encode(), 1)      # 1 = verbose
            print(f"[GraphedStep] hipGraphDebugDotPrint -> {rc} ({dot})", flush=True)
            g.instantiate()
        self._tables_id = id(self.optimizer._tables)
        if self.hybrid:          # the gradient tensors the recording writes (its backward assigned them after zero_grad): static from here on
            self.grads = [(p, p.grad) for g_ in self.optimizer.param_groups for p in g_["params"] if p.grad is not None]
        self.graph = g

    def _replay(self, x, mask, target):
        if x is not self.sx:
            self.sx.copy_(x, non_blocking=True)
        if mask is not self.smask:
            self.smask.copy_(mask, non_blocking=True)
        if target is not self.starget:
            self.starget.copy_(target, non_blocking=True)
        if id(self.optimizer._tables) != self._tables_id:
            raise L.PassionHipError("GraphedStep: optimizer.load_state_dict() replaced the moment tensors this recording updates; build a "
                                    "new GraphedStep")
        if not self.hybrid and self.step_fn is None:
            # the recording re-packs the conv weight images BEHIND its optimizer passes (for the next replay's forward): a weight changed
            # from outside since (load_state_dict, an eager step, weight averaging) is re-packed here, eagerly -- a host-side version
            # check, no launch when nothing moved
            from . import ops
            ops.repack_stale()
            ops.recast_stale()                    # (the same for the bf16 operands of the token GEMMs)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())      # CPU generator: follows torch.manual_seed
        w = (ctypes.c_uint32 * 2)(seed & 0xffffffff, (seed >> 32) & 0xffffffff)
        L.check(L.load().passion_write_words(self.words.data_ptr(), 2, ctypes.cast(w, c_void_p), L.stream_ptr()), "passion_write_words")
        if not self.hybrid:
            self.optimizer.graph_tick(self.cap)
        self.graph.replay()
        self.replays += 1
        if not self.hybrid and self.step_fn is None:
            from . import ops
            ops.mark_packed_fresh()               # ... the replay has re-packed every image from the weights it wrote (graph_tick moved their versions)
            ops.mark_cast_fresh()
        if self.hybrid:
            for p, g_ in self.grads:          # (the previous tail left views of the reduced buckets there)
                p.grad = g_
            ptrain.train_step(self.model, self.optimizer, None, None, None, phase="tail", reducer=self.reducer)
        return self.loss

    def input_buffers(self):
        """-> (x, mask, target): the tensors the recording reads, or None before it exists.  A loader may write the next batch straight
        into them (the H2D copy's destination) and hand THEM to __call__: the staging copies of _replay (x: 4 B / voxel, the float64
        label volume: 32 B / voxel -- 0.1 ms per 128^3 step) are then skipped, as the eager step has none either"""
        return (self.sx, self.smask, self.starget) if self.graph is not None else None

    def __call__(self, x, mask, target):
        """one training iteration on (x, mask, target); returns the loss (a static tensor once the step is recorded: the next call
        overwrites it)"""
        L.require_gpu(x)
        sig = self._signature(x, mask, target)
        if self.sig is not None and sig != self.sig:
            raise L.PassionHipError(f"GraphedStep was built for {self.sig}, called with {sig}: build another one (or call train_step)")
        self.sig = sig
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=x.device)
        self.calls += 1
        if self.graph is None:
            if self.calls <= self.warm:
                return self._eager(x, mask, target)
            self._record(x, mask, target)          # records, does not execute: the replay below is this call's iteration
        return self._replay(x, mask, target)
