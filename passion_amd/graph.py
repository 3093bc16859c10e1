"""One training iteration as a single HIP graph.

The eager step enqueues ~2000 kernels from Python; at 128^3 the forward is bound by that enqueue rate, not by
the GPU.  GraphedTrainStep records weight re-packing + forward + loss assembly + backward + AdamW once
(torch.cuda.CUDAGraph = hipGraph on ROCm; every kernel of libpassion_hip.so is launched on torch's current
stream, so the capture sees them) and replays it per iteration with the batch copied into static buffers.
Single-process only: the multi-GPU path (passion_amd.parallel.GradReducer) stays eager.

What the capture relies on:
  * no host synchronisation anywhere in the step (train.assemble_loss / LossState keep everything on the device);
  * caches keyed on tensor versions are bypassed while capturing (ops.repack_stale, ops.target_to_cl), because a
    replay must redo that work for the new weights / the new batch;
  * the optimizer is built with capturable=True (train.make_optimizer(..., capturable=True)).
"""
import torch

from . import train as _train


class GraphedTrainStep:
    def __init__(self, model, optimizer, x, mask, target, temp=4.0, state=None, warmup=False, modal_num=None,
                 warmup_iters=3, prof_mask=0):
        assert x.is_cuda, "GraphedTrainStep needs the batch on the GPU"
        self.model, self.optimizer, self.state = model, optimizer, state
        self.temp, self.warmup, self.modal_num = temp, warmup, modal_num
        self.x, self.mask, self.target = x.clone(), mask.clone(), target.clone()
        for g in optimizer.param_groups:
            assert g.get("capturable", False), "build the optimizer with capturable=True"
        model.is_training = True
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the default stream (allocator, attributes, autotune)
            for _ in range(warmup_iters):
                optimizer.zero_grad(set_to_none=True)
                self._fwd_bwd()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # Gradients live OUTSIDE the graph's memory pool: they are the tensors the warm-up steps left in .grad, cleared by
        # a (captured) foreach-zero at the top of every replay and accumulated into in place.  Letting the capture allocate
        # them (zero_grad(set_to_none=True) before the capture, the usual recipe) was measured to corrupt four Linear bias
        # gradients from the third replay on (ROCm 7.2 / torch 2.10: their pool blocks get re-used).
        if prof_mask:            # hipEvent pairs around the conv launches become event-record nodes of the graph:
            from . import ops    # ops.prof_read(keep=True) after a replay returns that replay's kernel durations
            ops.prof_read()
            ops.prof_enable(prof_mask)
        with torch.cuda.graph(self.graph):
            optimizer.zero_grad(set_to_none=False)
            self.loss = self._fwd_bwd()
            optimizer.step()
        if prof_mask:
            ops.prof_enable(0)
        torch.cuda.synchronize()

    def _fwd_bwd(self):
        m = self.model
        outs = m(self.x, self.mask, target=self.target, temp=self.temp)
        loss = _train.assemble_loss(outs, self.target, self.mask, m.mask_type, m.use_passion, self.state, self.warmup,
                                    self.modal_num)
        loss.backward()
        return loss.detach()

    def __call__(self, x=None, mask=None, target=None):
        """copy the batch into the static buffers (None = keep the previous one) and replay; returns the loss tensor
        (device, overwritten by the next replay)"""
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if mask is not None:
            self.mask.copy_(mask, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        self.graph.replay()
        return self.loss
