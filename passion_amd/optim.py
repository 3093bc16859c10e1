"""AdamW(amsgrad) with the update of all parameters fused into a handful of HIP launches.

Same constructor, math and state layout ('step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq') as torch.optim.AdamW
(reference code/train.py:95-96 builds torch.optim.AdamW(..., amsgrad=True) and checkpoints its state_dict,
train.py:358-364), so optimizer checkpoints interchange.  Parameters that are not f32 CUDA tensors fall back to
torch's own functional AdamW for that group.
"""
from ctypes import c_int64, c_void_p

import torch

from . import _lib as L


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if group["amsgrad"]:
                        st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            ok = all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32
                     for p in ps)
            if not ok:
                raise L.PassionHipError("FusedAdamW needs contiguous f32 CUDA parameters and gradients")
            steps = {int(self.state[p]["step"]) for p in ps}
            assert len(steps) == 1, "parameters of one group must share the step count"
            step = steps.pop() + 1
            n = len(ps)
            arr = lambda ts: (c_void_p * n)(*[t.data_ptr() for t in ts])     # noqa: E731
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            ms = [self.state[p]["exp_avg"] for p in ps]
            vs = [self.state[p]["exp_avg_sq"] for p in ps]
            vm = [self.state[p]["max_exp_avg_sq"] for p in ps] if group["amsgrad"] else ms
            numel = (c_int64 * n)(*[p.numel() for p in ps])
            b1, b2 = group["betas"]
            L.check(lib.passion_adamw_step(n, arr(ps), arr(grads), arr(ms), arr(vs), arr(vm), numel, float(group["lr"]), float(b1),
                                           float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                           1 if group["amsgrad"] else 0, L.stream_ptr()), "passion_adamw_step")
            for p in ps:
                self.state[p]["step"] += 1
                # the kernel wrote the parameters behind autograd's back: bump their version counters, the packed-weight
                # caches (ops.pack_weight / repack_stale) and any saved-tensor checks key on them
                torch.autograd.graph.increment_version(p)
        return loss
