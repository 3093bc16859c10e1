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
        self._tables = {}
        self._stepped = set()        # data_ptr of the parameters step_ready() already updated in this iteration
        self._capture = None         # HIP-graph capture (passion_amd.graph): {"hyper": device f32 [4 slots x 4], "subs": [...]}

    @torch.no_grad()
    def step_ready(self):
        """update, on the CURRENT stream, every parameter that has a gradient now and was not updated yet in this iteration; the
        closing step() takes the rest.  Called from the backward (ops.BackwardMarkerFn) when only the encoders' backward is left:
        the AdamW pass over the decoder / transformer parameters (33 of 35.6 M) then runs under it instead of after it."""
        self._step_impl(partial=True)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._step_impl(partial=False)
        self._stepped.clear()
        return loss

    def _step_impl(self, partial):
        lib = L.load()
        host = L.host() if all(p.is_cuda for g in self.param_groups for p in g["params"]) else None
        for gi, group in enumerate(self.param_groups):
            allp = group["params"]
            if host is not None:         # one C++ pass over the parameters instead of a few hundred attribute reads
                pptr, gptr = host.param_grad_ptrs(allp)
                live = [i for i, q in enumerate(gptr) if q and pptr[i] not in self._stepped]
                ps = [allp[i] for i in live]
            else:
                ps = [p for p in allp if p.grad is not None]
                live, pptr, gptr = None, None, None
            if not ps:
                continue
            # everything that does not change from step to step (validation, pointer tables of the parameters and their
            # moments) is built once per set of live parameters: the step is host-bound, 288 parameters x a few dict
            # lookups and CPU-tensor increments were 1.2 ms of it
            key = tuple(live) if live is not None else tuple(id(p) for p in ps)
            cache = self._tables.get((gi, key))
            cur_pptr = [pptr[i] for i in live] if live is not None else [p.data_ptr() for p in ps]
            if cache is None or cache["key"] != key or cache["pptr"] != cur_pptr:
                self._sync_steps()      # the live counts are in the old table: bring the per-parameter tensors up to date first
                for p in ps:
                    st = self.state[p]
                    if len(st) == 0:
                        st["step"] = torch.tensor(0.0)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        if group["amsgrad"]:
                            st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
                    raise L.PassionHipError("FusedAdamW needs contiguous f32 CUDA parameters and gradients")
                # torch keeps one step count per parameter (bias correction): parameters whose first gradient arrived late
                # form their own sub-table with their own count -- one launch per distinct count (normally one)
                by_step = {}
                for i, p in enumerate(ps):
                    by_step.setdefault(int(self.state[p]["step"]), []).append(i)
                subs = []
                for step0, idx in sorted(by_step.items()):
                    sp = [ps[i] for i in idx]
                    n = len(sp)
                    arr = lambda ts: (c_void_p * n)(*[t.data_ptr() for t in ts])     # noqa: E731
                    ms = [self.state[p]["exp_avg"] for p in sp]
                    vs = [self.state[p]["exp_avg_sq"] for p in sp]
                    vm = [self.state[p]["max_exp_avg_sq"] for p in sp] if group["amsgrad"] else ms
                    subs.append({"idx": idx, "p": arr(sp), "m": arr(ms), "v": arr(vs), "vm": arr(vm),
                                 "numel": (c_int64 * n)(*[p.numel() for p in sp]), "g": (c_void_p * n)(), "step": step0,
                                 "step_tensors": [self.state[p]["step"] for p in sp]})
                # tables of another partition of the same parameters (a full step() after partial ones, or the reverse) are
                # stale from here on: their step counts must never be written back again
                pset = set(cur_pptr)
                for k2 in [k2 for k2, c2 in self._tables.items() if k2[0] == gi and not pset.isdisjoint(c2["pptr"])]:
                    del self._tables[k2]
                cache = {"key": key, "pptr": cur_pptr, "subs": subs}
                self._tables[(gi, key)] = cache
            b1, b2 = group["betas"]
            for sub in cache["subs"]:
                garr = sub["g"]
                if live is not None:         # checked and made contiguous by param_grad_ptrs
                    garr[:] = [gptr[live[i]] for i in sub["idx"]]
                else:
                    for j, i in enumerate(sub["idx"]):
                        p = ps[i]
                        g = p.grad
                        if g.dtype != torch.float32 or not g.is_cuda:
                            raise L.PassionHipError("FusedAdamW needs contiguous f32 CUDA parameters and gradients")
                        if not g.is_contiguous():
                            g = p.grad = g.contiguous()
                        garr[j] = g.data_ptr()
                if self._capture is not None:
                    # being recorded into a HIP graph: the step-dependent scalars (lr * wd, lr / bias correction 1, 1 / sqrt(bias
                    # correction 2)) are read from device memory by the kernel; graph_tick() counts the step and writes them
                    # before every replay
                    cap = self._capture
                    slot = len(cap["subs"])
                    if slot >= 4:
                        raise L.PassionHipError("FusedAdamW: more than four distinct step counts in one recorded step")
                    cap["subs"].append((sub, group))
                    L.check(lib.passion_adamw_step_dev(len(sub["idx"]), sub["p"], garr, sub["m"], sub["v"], sub["vm"], sub["numel"],
                                                       float(b1), float(b2), float(group["eps"]), 1 if group["amsgrad"] else 0,
                                                       cap["hyper"].data_ptr() + 16 * slot, L.stream_ptr()), "passion_adamw_step_dev")
                    continue
                sub["step"] += 1
                L.check(lib.passion_adamw_step(len(sub["idx"]), sub["p"], garr, sub["m"], sub["v"], sub["vm"], sub["numel"],
                                               float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                               float(group["weight_decay"]), sub["step"], 1 if group["amsgrad"] else 0,
                                               L.stream_ptr()), "passion_adamw_step")
            # the kernel wrote the parameters behind autograd's back: bump their version counters, the packed-weight
            # caches (ops.pack_weight / repack_stale) and any saved-tensor checks key on them
            torch.autograd.graph.increment_version(ps)
            if partial:
                self._stepped.update(cur_pptr)

    # ---- a recorded training step (passion_amd.graph.GraphedStep)
    def begin_capture(self, hyper):
        """hyper: device f32 tensor of >= 16 elements; until end_capture() the launches read their scalars from it"""
        self._capture = {"hyper": hyper, "subs": []}

    def end_capture(self):
        cap, self._capture = self._capture, None
        return cap

    def graph_tick(self, cap):
        """before a replay of the step recorded under `cap`: count the step of every launch in it and hand the kernels their
        scalars (one tiny launch on the current stream, values passed by value: nothing on the host has to outlive the call)"""
        import ctypes
        lib = L.load()
        words = (ctypes.c_float * 16)()
        out3 = (ctypes.c_float * 3)()
        for slot, (sub, group) in enumerate(cap["subs"]):
            sub["step"] += 1
            b1, b2 = group["betas"]
            L.check(lib.passion_adamw_hyper(float(group["lr"]), float(b1), float(b2), float(group["weight_decay"]), sub["step"],
                                            ctypes.cast(out3, c_void_p)), "passion_adamw_hyper")
            words[4 * slot:4 * slot + 3] = out3[:]
        L.check(lib.passion_write_words(cap["hyper"].data_ptr(), 4 * len(cap["subs"]), ctypes.cast(words, c_void_p), L.stream_ptr()),
                "passion_write_words")
        # the kernels write the parameters behind autograd's back (see _step_impl)
        torch.autograd.graph.increment_version([p for g in self.param_groups for p in g["params"]])

    def zero_grad(self, set_to_none=True):
        host = L.host() if set_to_none and all(p.is_cuda for g in self.param_groups for p in g["params"]) else None
        if host is None:
            return super().zero_grad(set_to_none)
        for group in self.param_groups:
            host.drop_grads(group["params"])

    def _sync_steps(self):
        """the per-parameter 'step' tensors of torch's state layout, brought up to date (they are only read by state_dict)"""
        for cache in self._tables.values():
            for sub in cache["subs"]:
                for t in sub["step_tensors"]:
                    t.fill_(float(sub["step"]))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}
