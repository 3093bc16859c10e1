"""Depth (slab) sharding of ONE volume over the ranks -- BASELINE.json configs[4], SURVEY.md 8(e) row 2.

The reference has no counterpart (a full 240x240x155 forward is not defined by it): this is the multi-GPU mode the
north_star asks for, held to "sharded == unsharded" on the same volume.  Every rank owns a contiguous slab of D planes of every
activation (dim 1 of the channels-last tensors) at every level; what couples the slabs:

  3x3x3 conv (stride 1)   one halo plane from each neighbour (HaloExtendFn: the extended slab in one launch, passion_halo_extend; the
                          planes travel point-to-point over RCCL, or by one all_gather per exchange where the step is recorded as a
                          HIP graph -- batch_isend_irecv cannot be captured on this stack), reflect / zero only at the global ends;
                          the conv runs over the extended slab and the two outer output planes are dropped.  Backward: the halo
                          planes' gradients go back and are added (passion_halo_extend_bwd).
  3x3x3 conv (stride 2)   one plane from the low neighbour only
  InstanceNorm            statistics: the ranks' (sum x, sum x^2) all-reduced once per tensor (global_stats); backward: the conv
                          backward stops after its dgrad (passion_conv3d_bwd_phase 1), the (sum g, sum g x_hat) are all-reduced,
                          the finish runs with 1 / V_global (phase 2); halo planes get only the rstd * g term (their owner adds
                          the mean-subtraction terms)
  trilinear x k           global align_corners coordinates along D (passion_upsample_window_fwd / bwd) on the extended slab
  transformers            level-5 token volumes are all-gathered along D and the token path runs replicated (8 planes at 128^3:
                          one per GPU); its output is sliced back.  Backward of the gather = all-reduce + slice
  losses                  per-class partial sums all-reduced before the loss is formed (one (B,16) all-reduce per loss term)
  weight gradients        every rank's dW covers its slab: summed over the ranks by parallel.GradReducer like data parallel

`sharded_step_loss` is the mmformer training forward (plain and PASSION) in this mode; tests/test_depth_*.py hold it against the
unsharded model and the imported reference's fixtures (world 2, gloo rehearsal transport on one GPU).  In bf16 on the GPU the four
encoders run as grouped launches over the stacked batch and decoder_sep once over the stack (GROUPED), as in the data-parallel step.
`sharded_train_step` is the whole iteration, `GraphedDepthStep` the same as ONE recorded HIP graph, collectives included.
Unmeasured on multi-GPU hardware.
"""
import os
from ctypes import byref, c_void_p

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops
from .ops import _p

GROUPED = os.environ.get("PASSION_DEPTH_GROUPED", "1") != "0"     # encoders grouped, decoder_sep batched (0: one modality at a time)


class DepthCtx:
    """rank r of `world` owns planes [r * D / world, (r + 1) * D / world) of every level (D divisible by world at every level it
    is split; below that the caller gathers)"""

    def __init__(self, rank=None, world=None, group=None, loopback=False, halo_by_gather=None):
        self.group = group
        self.loopback = loopback         # world 1: exchange() still goes through the transport, every plane sent to this rank itself
        # halo planes through ONE all_gather of every rank's (first, last) plane instead of grouped send / recv pairs: what a recorded
        # step needs on this stack -- batch_isend_irecv under stream capture segfaults (ROCm 7.2 / torch 2.10 / RCCL 2.26,
        # tools/rccl_capture_probe.py micro_p2p), all_reduce / all_gather record and replay.  world x the bytes of the point-to-point
        # form (planes are <= 1 MB); GraphedDepthStep turns it on, PASSION_DEPTH_HALO=gather|p2p forces either
        env = os.environ.get("PASSION_DEPTH_HALO")
        self.halo_by_gather = (env == "gather") if env in ("gather", "p2p") else bool(halo_by_gather)
        self.n_exchange = self.n_allreduce = 0          # collectives issued (per step: read and reset by the caller)
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"

    @property
    def lo(self):
        return self.rank - 1 if self.rank > 0 else None

    @property
    def hi(self):
        return self.rank + 1 if self.rank < self.world - 1 else None

    def all_reduce(self, t):
        # loopback (world 1 on the real transport): the SUM over one rank is the identity, the CALL is the one every rank of an N > 1
        # job issues -- what tests/test_rccl_world1_gpu.py and the recorded step (GraphedDepthStep) exercise on a one-GPU box
        if self.world > 1 or (self.loopback and dist.is_initialized()):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.n_allreduce += 1
        return t

    def exchange(self, to_lo, to_hi):
        """send `to_lo` / `to_hi` to the neighbours, return what they sent (None at the global ends).  Point-to-point; the gloo
        rehearsal transport cannot move device tensors, so there the planes travel through host memory."""
        if self.halo_by_gather and dist.is_initialized() and (self.world > 1 or self.loopback) and not (self.backend == "gloo" and to_lo.is_cuda):
            mine = torch.stack([to_lo, to_hi])                                  # (2, B', P)
            every = torch.empty((self.world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(every, mine, group=self.group)
            self.n_exchange += 1
            if self.world == 1:
                self.last_loopback = (every[0, 0], every[0, 1])
                return None, None
            # the low neighbour's LAST plane is my low halo, the high neighbour's FIRST plane my high halo
            return (every[self.rank - 1, 1] if self.lo is not None else None), (every[self.rank + 1, 0] if self.hi is not None else None)
        self.n_exchange += 1 if (self.world > 1 or self.loopback) else 0
        if self.world == 1 and self.loopback and dist.is_initialized():
            # the calls of the N > 1 path against the real transport on a one-GPU box: a grouped send / recv pair per direction,
            # peer = this rank; what comes back is what was sent (the caller's ends stay global ends: the planes are dropped)
            r_lo, r_hi = torch.empty_like(to_lo), torch.empty_like(to_hi)
            reqs = [dist.P2POp(dist.isend, to_lo, self.rank, self.group), dist.P2POp(dist.irecv, r_lo, self.rank, self.group),
                    dist.P2POp(dist.isend, to_hi, self.rank, self.group), dist.P2POp(dist.irecv, r_hi, self.rank, self.group)]
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
            self.last_loopback = (r_lo, r_hi)
            return None, None
        if self.world == 1:
            return None, None
        via_host = self.backend == "gloo" and to_lo.is_cuda
        dev = to_lo.device
        s_lo, s_hi = (to_lo.cpu(), to_hi.cpu()) if via_host else (to_lo, to_hi)
        r_lo = torch.empty_like(s_hi) if self.lo is not None else None
        r_hi = torch.empty_like(s_lo) if self.hi is not None else None
        reqs = []
        if self.lo is not None:
            reqs += [dist.P2POp(dist.isend, s_lo, self.lo, self.group), dist.P2POp(dist.irecv, r_lo, self.lo, self.group)]
        if self.hi is not None:
            reqs += [dist.P2POp(dist.isend, s_hi, self.hi, self.group), dist.P2POp(dist.irecv, r_hi, self.hi, self.group)]
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
        if via_host:
            r_lo = r_lo.to(dev) if r_lo is not None else None
            r_hi = r_hi.to(dev) if r_hi is not None else None
        return r_lo, r_hi


def _plane(x, d):
    """plane d of every sample of a (B', D, ...) tensor as a contiguous (B', P) buffer (passion_halo_pack)"""
    Bp, D = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    buf = torch.empty(Bp, P, dtype=x.dtype, device=x.device)
    if x.is_cuda:
        L.check(L.load().passion_halo_pack(_p(x), _p(buf), L.dtype_code(x.dtype), Bp, D, P, d, L.stream_ptr()), "passion_halo_pack")
    else:                       # CPU rehearsal of the exchange logic (tests/test_depth_cpu.py)
        buf.copy_(x[:, d].reshape(Bp, P))
    return buf


def _put_plane(buf, y, d, accumulate):
    Bp, D = y.shape[0], y.shape[1]
    P = y[0, 0].numel()
    if y.is_cuda:
        L.check(L.load().passion_halo_unpack(_p(buf), _p(y), L.dtype_code(y.dtype), Bp, D, P, d, 1 if accumulate else 0, L.stream_ptr()),
                "passion_halo_unpack")
    elif accumulate:
        y[:, d] += buf.view(y[:, d].shape)
    else:
        y[:, d] = buf.view(y[:, d].shape)


def _exchange_active(shard):
    return shard.world > 1 or (shard.loopback and dist.is_initialized())


class HaloExtendFn(torch.autograd.Function):
    """(B', Dl, ...) slab -> (B', n_lo + Dl + n_hi, ...): n_hi in {0, 1} planes from the high neighbour, n_lo in {0, 1, 2} from the
    low one (2 = one zero plane in front of the halo plane: the stride-2 convs read planes 2j-1 .. 2j+1; 0 / n_hi = 0 at a global
    end of a ZERO-padded pre-norm conv: its padding is zero AFTER norm + activation, so there the conv's own padding is used and
    no plane is added).  At the global ends of the other ops the halo plane is reflect (plane 1 / Dl-2) or zeros.
    hi_used: the op reads a high halo at all (stride-1 convs, upsampling) -- tells the backward whether the low neighbour sent a
    gradient for my first plane.
    GPU tensors (round 5): the extended slab is ONE launch (passion_halo_extend: owned planes, both halo planes from their sources,
    zero planes), its adjoint one more (passion_halo_extend_bwd: owned planes + reflect folds + the two send planes); the first
    form's torch.zeros / copy_ / plane packs / plane puts were ~13 launches per conv and direction."""

    @staticmethod
    def forward(ctx, x, shard, reflect, n_lo, n_hi, hi_used=True):
        x = x.contiguous()
        Dl = x.shape[1]
        ctx.shard, ctx.reflect, ctx.n_lo, ctx.n_hi, ctx.Dl, ctx.hi_used = shard, reflect, n_lo, n_hi, Dl, hi_used
        ctx.fast = x.is_cuda and Dl > 1 and (x[0, 0].numel() * x.element_size()) % 16 == 0
        if ctx.fast:
            Bp, P = x.shape[0], x[0, 0].numel()
            r_lo = r_hi = None
            if _exchange_active(shard):
                r_lo, r_hi = shard.exchange(_plane(x, 0), _plane(x, Dl - 1))
            ext = torch.empty((Bp, n_lo + Dl + n_hi) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
            lo_mode = 0 if not n_lo else (1 if r_lo is not None else (2 if reflect else 0))
            hi_mode = 0 if not n_hi else (1 if r_hi is not None else (2 if reflect else 0))
            L.check(L.load().passion_halo_extend(_p(x), _p(r_lo) if lo_mode == 1 else c_void_p(0), _p(r_hi) if hi_mode == 1 else c_void_p(0), _p(ext),
                                                 L.dtype_code(x.dtype), Bp, Dl, P, n_lo, n_hi, lo_mode, hi_mode, 1, Dl - 2, L.stream_ptr()),
                    "passion_halo_extend")
            return ext
        ext = torch.zeros((x.shape[0], n_lo + Dl + n_hi) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        ext[:, n_lo:n_lo + Dl].copy_(x)
        r_lo, r_hi = shard.exchange(_plane(x, 0), _plane(x, Dl - 1))       # my first plane goes down, my last plane goes up
        # reflect at a global end mirrors plane 1 (resp. D-2) of the VOLUME: with one plane per rank (level 5 of a 128^3 volume on
        # 8 GPUs) that is the neighbour's plane, i.e. the halo plane received from the other side
        if n_lo:
            if r_lo is not None:
                _put_plane(r_lo, ext, n_lo - 1, False)
            elif reflect:
                _put_plane(_plane(x, 1) if Dl > 1 else r_hi, ext, n_lo - 1, False)
        if n_hi:
            if r_hi is not None:
                _put_plane(r_hi, ext, n_lo + Dl, False)
            elif reflect:
                _put_plane(_plane(x, Dl - 2) if Dl > 1 else r_lo, ext, n_lo + Dl, False)
        return ext

    @staticmethod
    def backward(ctx, g):
        shard, n_lo, n_hi, Dl = ctx.shard, ctx.n_lo, ctx.n_hi, ctx.Dl
        g = g.contiguous()
        if ctx.fast:
            Bp, P = g.shape[0], g[0, 0].numel()
            gx = torch.empty((Bp, Dl) + tuple(g.shape[2:]), dtype=g.dtype, device=g.device)
            active = _exchange_active(shard)
            send = torch.empty(2, Bp, P, dtype=g.dtype, device=g.device) if active else None
            # a halo plane that was a reflection of my own plane 1 / Dl - 2 (global end): its gradient folds onto that plane
            fold_lo = 1 if (ctx.reflect and n_lo and shard.lo is None) else -1
            fold_hi = Dl - 2 if (ctx.reflect and n_hi and shard.hi is None) else -1
            L.check(L.load().passion_halo_extend_bwd(_p(g), _p(gx), _p(send) if active else c_void_p(0), L.dtype_code(g.dtype), Bp, Dl, P, n_lo, n_hi,
                                                     fold_lo, fold_hi, L.stream_ptr()), "passion_halo_extend_bwd")
            if active:
                r_lo, r_hi = shard.exchange(send[0], send[1])
                if r_hi is not None:                  # the high neighbour's low halo was my last plane
                    _put_plane(r_hi, gx, Dl - 1, True)
                if r_lo is not None and ctx.hi_used:  # the low neighbour's high halo was my first plane
                    _put_plane(r_lo, gx, 0, True)
            return gx, None, None, None, None, None
        gx = g[:, n_lo:n_lo + Dl].contiguous()
        # gradient of the halo planes: back to their owners (the low halo is the low neighbour's LAST plane)
        zero = torch.zeros(g.shape[0], g[0, 0].numel(), dtype=g.dtype, device=g.device)
        g_lo = _plane(g, n_lo - 1) if n_lo else zero
        g_hi = _plane(g, n_lo + Dl) if n_hi else zero
        if ctx.reflect and Dl == 1:               # the mirrored plane was the neighbour's (see forward): its gradient travels with the halo's
            if shard.lo is None and n_lo:
                g_hi = g_hi + g_lo
            if shard.hi is None and n_hi:
                g_lo = g_lo + g_hi
        r_lo, r_hi = shard.exchange(g_lo, g_hi)
        if r_hi is not None:                      # the high neighbour's low halo was my last plane
            _put_plane(r_hi, gx, Dl - 1, True)
        if r_lo is not None and ctx.hi_used:      # the low neighbour's high halo was my first plane
            _put_plane(r_lo, gx, 0, True)
        if ctx.reflect and Dl > 1:
            if shard.lo is None and n_lo:
                _put_plane(g_lo, gx, 1, True)
            if shard.hi is None and n_hi:
                _put_plane(g_hi, gx, Dl - 2, True)
        return gx, None, None, None, None, None


class CropDFn(torch.autograd.Function):
    """planes [o0, o0 + n) of a (B', D, ...) tensor (the output planes of an extended slab this rank owns); backward: the gradient
    with zero planes around it in ONE launch (autograd's slice backward is zeros + copy_: two launches, twice the bytes)"""

    @staticmethod
    def forward(ctx, y, o0, n):
        ctx.o0, ctx.n, ctx.D = o0, n, y.shape[1]
        return y[:, o0:o0 + n].contiguous()       # one sample: a view

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        Bp, P = g.shape[0], g[0, 0].numel()
        n_lo, n_hi = ctx.o0, ctx.D - ctx.o0 - ctx.n
        if not g.is_cuda or (P * g.element_size()) % 16:
            out = torch.zeros((Bp, ctx.D) + tuple(g.shape[2:]), dtype=g.dtype, device=g.device)
            out[:, ctx.o0:ctx.o0 + ctx.n].copy_(g)
            return out, None, None
        out = torch.empty((Bp, ctx.D) + tuple(g.shape[2:]), dtype=g.dtype, device=g.device)
        L.check(L.load().passion_halo_extend(_p(g), c_void_p(0), c_void_p(0), _p(out), L.dtype_code(g.dtype), Bp, ctx.n, P, n_lo, n_hi, 0, 0, 0, 0,
                                             L.stream_ptr()), "passion_halo_extend")
        return out, None, None


class GatherDFn(torch.autograd.Function):
    """all-gather along dim 1 (D); backward: the ranks' gradients of the gathered tensor summed, own slab returned"""

    @staticmethod
    def forward(ctx, x, shard):
        ctx.shard, ctx.Dl = shard, x.shape[1]
        if shard.world == 1:
            return x
        parts = [torch.empty_like(x) for _ in range(shard.world)]
        dist.all_gather(parts, x.contiguous(), group=shard.group)
        return torch.cat(parts, 1)

    @staticmethod
    def backward(ctx, g):
        if ctx.shard.world == 1:
            return g, None
        g = ctx.shard.all_reduce(g.contiguous().clone())
        return g[:, ctx.shard.rank * ctx.Dl:(ctx.shard.rank + 1) * ctx.Dl].contiguous(), None


def global_stats(x, shard, eps=1e-5):
    """InstanceNorm (mean, rstd) per (b, c) over the WHOLE volume from the ranks' slabs: local statistics kernel, then one
    all-reduce of (n, n mean, n (var + mean^2)) in double -- SURVEY 8e "per IN: all-reduce of (sum x, sum x^2) * C floats".
    GPU: two tiny launches around the all-reduce (passion_stats_to_moments / passion_moments_to_stats); a single rank without a
    transport has nothing to combine."""
    st = ops.instnorm_stats(x.detach(), eps)                                # (B, C, 2) = mean, rstd of the slab
    if not _exchange_active(shard):
        return st
    n = float(x[0, ..., 0].numel())                                         # voxels of THIS rank's slab (slabs may differ in size)
    if x.is_cuda:
        BC = st.shape[0] * st.shape[1]
        mom = torch.empty(st.shape[0], st.shape[1], 3, dtype=torch.float64, device=x.device)
        L.check(L.load().passion_stats_to_moments(_p(st), _p(mom), BC, n, float(eps), L.stream_ptr()), "passion_stats_to_moments")
        shard.all_reduce(mom)
        out = torch.empty_like(st)
        L.check(L.load().passion_moments_to_stats(_p(mom), _p(out), BC, float(eps), L.stream_ptr()), "passion_moments_to_stats")
        return out
    st = st.double()
    mean, var = st[..., 0], (st[..., 1].pow(-2) - eps).clamp_min(0)
    # parallel-variance combination in double (Chan et al.): all-reduce (n, n * mean, M2 + n * mean^2); the between-rank term
    # sum n_r (mean_r - mean_g)^2 is formed from the reduced moments, so only the ranks' DIFFERENCES of means meet in it
    mom = torch.stack([torch.full_like(mean, n), mean * n, (var + mean * mean) * n], -1).contiguous()
    shard.all_reduce(mom)
    ng = mom[..., 0]
    gm = mom[..., 1] / ng
    gv = (mom[..., 2] / ng - gm * gm).clamp_min(0)
    return torch.stack([gm, (gv + eps).rsqrt()], -1).float().contiguous()


class ShardedConvFn(torch.autograd.Function):
    """ops.ConvBlockFn over (extended) slabs: same forward launch; the backward is split around the all-reduce of the
    InstanceNorm-backward sums (passion_conv3d_bwd_phase)"""

    @staticmethod
    def forward(ctx, cfg, shard, win, Dg, weight, bias, packed, *xs):
        L.require_gpu(weight, *xs)
        lib = L.load()
        Cout = weight.shape[0]
        pad = cfg.ksize // 2
        Do = (cfg.D + 2 * pad - cfg.ksize) // cfg.stride + 1
        Ho = (cfg.H + 2 * pad - cfg.ksize) // cfg.stride + 1
        Wo = (cfg.W + 2 * pad - cfg.ksize) // cfg.stride + 1
        y = torch.empty(cfg.B, Do, Ho, Wo, Cout, dtype=cfg.out_dtype, device=weight.device)
        d = ops._fill_desc(cfg, xs)
        d.Cout = Cout
        b32 = bias.detach().float() if bias is not None else None
        L.check(lib.passion_conv3d_fwd(byref(d), _p(packed), _p(b32), c_void_p(0), _p(y), L.stream_ptr()), "passion_conv3d_fwd")
        ctx.cfg, ctx.packed, ctx.has_bias, ctx.wshape = cfg, packed, bias is not None, weight.shape
        ctx.shard, ctx.win, ctx.Dg = shard, win, Dg
        ctx.save_for_backward(*xs)
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg, xs, shard = ctx.cfg, ctx.saved_tensors, ctx.shard
        lib = L.load()
        dy = dy.contiguous()
        dev = dy.device
        nW = ctx.wshape.numel()
        buf = torch.empty(nW + ctx.wshape[0], dtype=torch.float32, device=dev)
        dW, db = buf[:nW].view(ctx.wshape), buf[nW:]
        dxs = [torch.empty_like(x) if ctx.needs_input_grad[7 + i] else None for i, x in enumerate(xs)]
        d = ops._fill_desc(cfg, xs, dxs)
        d.Cout = ctx.wshape[0]
        Cin = sum(s.C for s in cfg.segs)
        scratch = torch.empty(lib.passion_conv_bwd_scratch_floats(byref(d)), dtype=torch.float32, device=dev)
        vscale = float(cfg.D) / float(ctx.Dg)                 # V of the (extended) slab / V of the whole volume
        args = (byref(d), _p(ctx.packed), _p(dy), _p(dW), _p(db) if ctx.has_bias else c_void_p(0), _p(scratch))
        L.check(lib.passion_conv3d_bwd_phase(*args, 1, vscale, ctx.win[0], ctx.win[1], L.stream_ptr()), "passion_conv3d_bwd_phase 1")
        if any(dx is not None for dx in dxs):
            shard.all_reduce(scratch[:2 * cfg.B * Cin])       # (sum g, sum g x_hat) of every rank's slab
            L.check(lib.passion_conv3d_bwd_phase(*args, 2, vscale, ctx.win[0], ctx.win[1], L.stream_ptr()), "passion_conv3d_bwd_phase 2")
        return (None, None, None, None, dW, db if ctx.has_bias else None, None, *dxs)


class ShardedConvGroupedFn(torch.autograd.Function):
    """ShardedConvFn for G weight sets over one stacked batch (sample s uses set s // (B / G)): the four modality encoders as ONE
    launch per layer and phase (passion_conv3d_fwd_grouped / passion_conv3d_bwd_grouped_phase), one halo exchange, one statistics
    all-reduce and one all-reduce of the InstanceNorm-backward sums instead of four of each"""

    @staticmethod
    def forward(ctx, cfg, shard, win, Dg, G, packed, x, *wb):
        ws, bs = wb[:G], wb[G:]
        L.require_gpu(ws[0], x)
        lib = L.load()
        Cout = ws[0].shape[0]
        pad = cfg.ksize // 2
        Do = (cfg.D + 2 * pad - cfg.ksize) // cfg.stride + 1
        Ho = (cfg.H + 2 * pad - cfg.ksize) // cfg.stride + 1
        Wo = (cfg.W + 2 * pad - cfg.ksize) // cfg.stride + 1
        y = torch.empty(cfg.B, Do, Ho, Wo, Cout, dtype=cfg.out_dtype, device=x.device)
        d = ops._fill_desc(cfg, (x,))
        d.Cout = Cout
        n4 = packed.numel() // G
        pk = (c_void_p * G)(*[packed.data_ptr() + 4 * g * n4 for g in range(G)])
        b32 = [b.detach().float().contiguous() for b in bs]
        bp = (c_void_p * G)(*[b.data_ptr() for b in b32])
        L.check(lib.passion_conv3d_fwd_grouped(byref(d), G, pk, bp, c_void_p(0), _p(y), c_void_p(0), 1e-5, L.stream_ptr()),
                "passion_conv3d_fwd_grouped")
        ctx.cfg, ctx.packed, ctx.G, ctx.wshape = cfg, packed, G, ws[0].shape
        ctx.shard, ctx.win, ctx.Dg = shard, win, Dg
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg, (x,), shard, G = ctx.cfg, ctx.saved_tensors, ctx.shard, ctx.G
        lib = L.load()
        dy = dy.contiguous()
        dev = dy.device
        nW, Cout = ctx.wshape.numel(), ctx.wshape[0]
        buf = torch.empty(G, nW + Cout, dtype=torch.float32, device=dev)
        dx = torch.empty_like(x) if ctx.needs_input_grad[6] else None
        d = ops._fill_desc(cfg, (x,), [dx])
        d.Cout = Cout
        Cin = sum(s.C for s in cfg.segs)
        scratch = torch.empty(lib.passion_conv_bwd_scratch_floats(byref(d)), dtype=torch.float32, device=dev)
        vscale = float(cfg.D) / float(ctx.Dg)
        n4 = ctx.packed.numel() // G
        pk = (c_void_p * G)(*[ctx.packed.data_ptr() + 4 * g * n4 for g in range(G)])
        dwp = (c_void_p * G)(*[buf[g].data_ptr() for g in range(G)])
        args = (byref(d), G, pk, _p(dy), dwp, _p(scratch))
        L.check(lib.passion_conv3d_bwd_grouped_phase(*args, 1, vscale, ctx.win[0], ctx.win[1], L.stream_ptr()), "passion_conv3d_bwd_grouped_phase 1")
        if dx is not None:
            shard.all_reduce(scratch[:2 * cfg.B * Cin])
            L.check(lib.passion_conv3d_bwd_grouped_phase(*args, 2, vscale, ctx.win[0], ctx.win[1], L.stream_ptr()), "passion_conv3d_bwd_grouped_phase 2")
        return (None, None, None, None, None, None, dx, *[buf[g, :nW].view(ctx.wshape) for g in range(G)], *[buf[g, nW:] for g in range(G)])


class WindowUpsampleFn(torch.autograd.Function):
    """trilinear x k (align_corners=True) of an extended slab (planes [id0, id0 + Dl_in) of Dg) onto output planes
    [od0, od0 + Dl_out) of k * Dg"""

    @staticmethod
    def forward(ctx, x, k, id0, Dg, Dl_out, od0):
        x = x.contiguous()
        B, Dl_in, H, W, C = x.shape
        y = torch.empty(B, Dl_out, H * k, W * k, C, dtype=x.dtype, device=x.device)
        L.check(L.load().passion_upsample_window_fwd(_p(x), _p(y), L.dtype_code(x.dtype), B, Dl_in, id0, Dg, Dl_out, od0, H, W, C, k,
                                                     L.stream_ptr()), "passion_upsample_window_fwd")
        ctx.meta = (x.shape, k, id0, Dg, Dl_out, od0)
        return y

    @staticmethod
    def backward(ctx, dy):
        (B, Dl_in, H, W, C), k, id0, Dg, Dl_out, od0 = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty(B, Dl_in, H, W, C, dtype=dy.dtype, device=dy.device)
        lib = L.load()
        if k > 2:           # the deep-supervision maps: three separable passes (the one-pass gather walks (2k + 1)^3 candidates per voxel)
            scratch = torch.empty(lib.passion_upsample_window_bwd_scratch_floats(B, Dl_out, H, W, C, k), dtype=torch.float32, device=dy.device)
            L.check(lib.passion_upsample_window_bwd_sep(_p(dy), _p(dx), L.dtype_code(dy.dtype), B, Dl_in, id0, Dg, Dl_out, od0, H, W, C, k,
                                                        _p(scratch), L.stream_ptr()), "passion_upsample_window_bwd_sep")
            return dx, None, None, None, None, None
        L.check(lib.passion_upsample_window_bwd(_p(dy), _p(dx), L.dtype_code(dy.dtype), B, Dl_in, id0, Dg, Dl_out, od0, H, W, C, k,
                                                L.stream_ptr()), "passion_upsample_window_bwd")
        return dx, None, None, None, None, None


class ShardedWceDiceFn(torch.autograd.Function):
    """softmax_weighted_loss_bs + dice_loss_bs (criterions.py:59-76, 25-38) of a slab of probabilities: the sixteen per-class sums
    of the loss kernel are all-reduced before the two scalars are formed; the backward kernel takes the global sums"""

    @staticmethod
    def forward(ctx, probs, target_cl, shard, layout):
        probs = probs.contiguous()
        B, V, C, ps = ops._strides(probs, layout)
        lib = L.load()
        scratch = torch.empty(lib.passion_wce_dice_sums_floats(B), dtype=torch.float32, device=probs.device)
        out = torch.empty(B, 2, dtype=torch.float32, device=probs.device)
        L.check(lib.passion_wce_dice_fwd(_p(probs), *ps, _p(target_cl), B, target_cl.shape[0], V, C, _p(scratch), _p(out), L.stream_ptr()), "passion_wce_dice_fwd")
        sums = scratch[:B * 16].view(B, 16).clone()    # the sixteen sums; the rest of the scratch is per-block reduction slots
        shard.all_reduce(sums)
        Vg = float(V * shard.world)
        s = sums.double()
        rt = s[:, 8:12].sum(1, keepdim=True)
        w = 1.0 - s[:, 8:12] / rt
        ce = (-(w * s[:, 12:16]).sum(1)) / Vg                                   # wce_dice_final_kernel with the global sums
        dice = 1.0 - (2.0 * s[:, 0:4] / (s[:, 4:8] + s[:, 8:12] + 1e-7)).sum(1) / 4.0
        ctx.layout, ctx.scale = layout, 1.0 / shard.world                        # the kernel divides by the local V
        ctx.save_for_backward(probs, target_cl, sums)
        return torch.stack([ce, dice], 1).float()

    @staticmethod
    def backward(ctx, g):
        probs, target_cl, sums = ctx.saved_tensors
        B, V, C, ps = ops._strides(probs, ctx.layout)
        g = g.contiguous().float().clone()
        g[:, 0] *= ctx.scale
        dp = torch.empty_like(probs)
        L.check(L.load().passion_wce_dice_bwd(_p(probs), *ps, _p(target_cl), _p(sums), _p(g), _p(dp), B, target_cl.shape[0], V, C, L.stream_ptr()),
                "passion_wce_dice_bwd")
        return dp, None, None, None


class AllReduceSumFn(torch.autograd.Function):
    """sum over the ranks of a small tensor of partial sums.  Every rank differentiates its own copy of the (global) loss and the
    parameter gradients are summed over the ranks afterwards, so what the backward must be depends on what consumes the sum:
      reduce_grad=False  the sum feeds a computation every rank repeats identically (the last reduction of a loss term): the
                         gradient with respect to the local partial sum is the incoming gradient itself;
      reduce_grad=True   the sum feeds RANK-LOCAL work whose results are reduced again later (the global prototypes, which every
                         rank's similarity maps depend on): the gradient is the sum of all ranks' incoming gradients -- another
                         all-reduce.  (With the identity here only the terms 'my maps x my features' would survive.)"""

    @staticmethod
    def forward(ctx, t, shard, reduce_grad=False):
        ctx.shard, ctx.reduce_grad = shard, reduce_grad
        out = t.detach().clone().contiguous()
        shard.all_reduce(out)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.reduce_grad:
            g = g.contiguous().clone()
            ctx.shard.all_reduce(g)
        return g, None, None


def sharded_kl(ps, pt, temp, shard):
    """temp_kl_loss_bs (criterions.py:92-103) of slabs of already softmaxed (+ upsampled) probabilities: the kernel's mean over the
    local voxels, averaged over the (equal) slabs"""
    return AllReduceSumFn.apply(ops.kl_probs(ps, pt, temp), shard) / shard.world


def sharded_proto(fs, ft, target_cl, shard):
    """prototype_passion_loss_bs (criterions.py:144-180) over slabs, TWO rounds of reductions: (1) the per-class voxel
    counts and feature sums -> global prototypes (and the global decision which classes every sample holds, :157); (2) the sums of
    the squared / absolute differences of the student's and the teacher's cosine-similarity maps.  fs (B,Dl,H,W,Cf) student
    features (gradient), ft teacher features (detached), target_cl (B,Dl,H,W,4) f32 one-hot.  -> ((B,1) loss, (B,1) dist).
    torch elementwise ops on the slab: this mode is bound by its exchanges, not by these passes."""
    eps = 1e-5
    fs32, ft32 = fs.float(), ft.detach().float()
    B = fs.shape[0]
    tc = target_cl.reshape(B, -1, 4)                                            # (B, V, 4)
    f_s, f_t = fs32.reshape(B, -1, fs.shape[-1]), ft32.reshape(B, -1, ft.shape[-1])
    cnt = AllReduceSumFn.apply(tc.sum(1), shard)                                # round 1: (B, 4) voxel counts ...
    sums = AllReduceSumFn.apply(torch.cat([torch.einsum("bvk,bvc->bkc", tc, f_s), torch.einsum("bvk,bvc->bkc", tc, f_t)], 2), shard, True)
    Cf = fs.shape[-1]
    keep = (cnt > 0).all(0)                                                     # a class takes part only if EVERY sample has it (:157)
    proto_s = sums[..., :Cf] / (cnt.unsqueeze(-1) + eps)                        # (B, 4, Cf)
    proto_t = sums[..., Cf:] / (cnt.unsqueeze(-1) + eps)

    def cos(f, pr):                                                             # F.cosine_similarity(dim = channel, eps): (B, V, 4)
        num = torch.einsum("bvc,bkc->bvk", f, pr)
        return num / (f.norm(dim=2, keepdim=True).clamp_min(eps) * pr.norm(dim=2).unsqueeze(1).clamp_min(eps))

    diff = (cos(f_s, proto_s) - cos(f_t, proto_t))[..., keep]
    part = torch.stack([(diff * diff).sum((1, 2)), diff.abs().sum((1, 2))], 1)  # round 2: (B, 2) local sums
    tot = AllReduceSumFn.apply(part, shard)
    n = float(f_s.shape[1] * shard.world) * float(keep.sum())                   # mean over (classes kept, global voxels)
    return (tot[:, 0:1] / n), (tot[:, 1:2] / n).detach()


# ------------------------------------------------------------------------------------------------------------------
# the mmformer training forward on slabs
# ------------------------------------------------------------------------------------------------------------------
class Slab:
    """a slab (B, Dl, H, W, C) + its global InstanceNorm statistics (computed once, shared by the consumers) + modality mask"""

    def __init__(self, t, shard, mask=None, col=0, part_of=None):
        self.t, self.shard, self.mask, self.col, self._st = t, shard, mask, col, None
        self._part_of = part_of                               # (stacked Slab, first sample, samples): statistics = rows of the stack's

    def stats(self):
        if self._st is None:
            if self._part_of is not None:
                whole, b0, nb = self._part_of
                self._st = whole.stats()[b0:b0 + nb]
            else:
                self._st = global_stats(self.t, self.shard)
        return self._st

    def with_mask(self, mask, col):
        s = Slab(self.t, self.shard, mask, col)
        s._st = self.stats()
        return s


def conv(mod, feats, shard, Dg, norm=True, residual=None, out_dtype=None):
    """one conv layer (nn.Conv3d holder `mod`) over the virtual concat of the slabs `feats`; Dg = depth of the whole volume at
    the conv's INPUT level"""
    if not isinstance(feats, (list, tuple)):
        feats = [feats]
    from .models.blocks import PAD
    mods = None
    if isinstance(mod, (list, tuple)):                        # G weight sets over one stacked batch (ShardedConvGroupedFn)
        mods, mod = list(mod), mod[0]
        assert len(feats) == 1
    k, stride = mod.kernel_size[0], mod.stride[0]
    reflect = mod.padding_mode == "reflect"
    Dl = feats[0].t.shape[1]
    n_lo, n_hi = (0, 0) if k == 1 else ((2, 0) if stride == 2 else (1, 1))
    if k == 3 and not reflect:                                # zero padding of a pre-norm conv: the conv's own, at the global ends
        assert stride == 1
        n_lo, n_hi = (0 if shard.lo is None else 1), (0 if shard.hi is None else 1)
    segs = []
    for f in feats:
        t = HaloExtendFn.apply(f.t, shard, reflect, n_lo, n_hi, stride == 1) if k == 3 else f.t
        segs.append(ops.SegIn(t, stats=f.stats() if norm else None, mask=f.mask, mask_col=f.col))
    cfg = ops._ConvCfg()
    x0 = segs[0].x
    cfg.segs, cfg.D, cfg.H, cfg.W = segs, x0.shape[1], x0.shape[2], x0.shape[3]
    cfg.ksize, cfg.stride, cfg.pad_mode, cfg.slope = k, stride, PAD[mod.padding_mode], 0.2
    cfg.B, cfg.out = x0.shape[0], None
    cfg.out_dtype = out_dtype if out_dtype is not None else x0.dtype
    if mods is None:
        cache = mod.__dict__.setdefault("_passion_pack", {})
        packed = ops.pack_weight(mod.weight, cache)
    win = (n_lo, n_lo + Dl)                                   # the planes of the extended slab this rank owns
    Dg_ext = Dg                                               # 1 / V_global = (D_ext / Dg) / V_ext
    if mods is not None:
        gp = ops._group_pack([m.weight for m in mods])
        y = ShardedConvGroupedFn.apply(cfg, shard, win, Dg_ext, len(mods), gp, segs[0].x, *[m.weight for m in mods], *[m.bias for m in mods])
    else:
        y = ShardedConvFn.apply(cfg, shard, win, Dg_ext, mod.weight, mod.bias, packed, *[s.x for s in segs])
    if k == 3:
        o0 = 1 if stride == 2 else n_lo
        if y.shape[1] != Dl // stride:
            y = CropDFn.apply(y, o0, Dl // stride)            # drop the output planes that belong to the neighbours
    if residual is not None:
        y = y + residual
    return Slab(y, shard)


def up(f, k, shard, Dg):
    """trilinear x k of a slab whose whole volume has Dg planes"""
    Dl = f.t.shape[1]
    ext = HaloExtendFn.apply(f.t, shard, False, 1, 1)
    d0 = shard.rank * Dl
    return Slab(WindowUpsampleFn.apply(ext, k, d0 - 1, Dg, Dl * k, d0 * k), shard, f.mask, f.col)


def seg_pair(probs_cl, target_cl, shard):
    return ShardedWceDiceFn.apply(probs_cl, target_cl, shard, "cl")


def sharded_step_loss(model, shard, x, mask, target, Dg, temp=4.0, state=None):
    """mmformer.Model.forward (training; reference mmformer.py:381-465,533-561,637-657, and with model.use_passion the four uni-modal
    students + KL / prototype losses of :468-633) + train.py's loss assembly (:413-440 plain, :228-280 PASSION) for ONE volume
    whose D axis is split over the ranks.
    x (B, 4, Dl, H, W), target (B, 4, Dl, H, W): this rank's slabs; Dg: depth of the whole volume.
    -> (loss, pred_local (B, 4, Dl, H, W), prm_loss (B, 1), sep_loss (B, 4)[, kl_loss, proto_loss, dist (B, 4) each]); the losses are
    global (equal on every rank)"""
    from .models.mmformer import MODALS
    B, _, Dl, H, W = x.shape
    dt = model.compute_dtype
    idt = model.mask_type != "pdt"
    mask = mask.to(device=x.device, dtype=torch.bool).contiguous()
    ops.repack_stale()
    ops.recast_stale()
    xin = x.to(dt).contiguous()
    xe = HaloExtendFn.apply(xin.view(B * 4, Dl, H, W), shard, True, 1, 1).view(B, 4, Dl + 2, H, W)
    vol = (Dl + 2) * H * W
    # the four encoders as ONE grouped launch per layer over the stacked batch [flair | t1ce | t1 | t2] (as models/mmformer.py
    # encoders_grouped): a quarter of the launches, halo exchanges and all-reduces; bf16 on the GPU (the grouped kernels), else one by one
    grouped = GROUPED and x.is_cuda and dt == torch.bfloat16 and L.host() is not None
    enc_mods = [getattr(model, f"{m}_encoder") for m in MODALS]
    firsts = []
    for i, enc in enumerate(enc_mods):                        # first layer: the raw input, one channel (no grouped weight gradient)
        seg = ops.SegIn(xe, mask=mask if idt else None, mask_col=i, C=1, offset=i * vol, bstride=4 * vol)
        cache = enc.e1_c1.__dict__.setdefault("_passion_pack", {})
        y = ops.conv_block([seg], enc.e1_c1.weight, enc.e1_c1.bias, Dl + 2, H, W, ksize=3, stride=1, pad_mode=L.PAD_REFLECT,
                           out_dtype=dt, pack_cache=cache)
        firsts.append(y[:, 1:1 + Dl])

    def encoder_levels(f, pick):
        feats, dg = [], Dg
        for lvl in range(1, 6):
            if lvl > 1:
                f = conv(pick(f"e{lvl}_c1"), f, shard, dg)
                dg //= 2
            h = conv(pick(f"e{lvl}_c2"), f, shard, dg)
            f = conv(pick(f"e{lvl}_c3"), h, shard, dg, residual=f.t)
            feats.append(f)
        return feats

    stk = None
    if grouped:
        stk = encoder_levels(Slab(torch.cat(firsts, 0), shard), lambda name: [getattr(e, name).conv for e in enc_mods])
        encs = [[None] * 5 for _ in range(4)]
        for l, f in enumerate(stk):
            for i, part in enumerate(ops.unstack(f.t, 4)):
                encs[i][l] = Slab(part, shard, part_of=(f, i * B, B))
    else:
        encs = [encoder_levels(Slab(firsts[i].contiguous(), shard), lambda name, e=enc: getattr(e, name).conv)
                for i, enc in enumerate(enc_mods)]
    dg5 = Dg // 16
    # ---- token path: gathered along D, replicated
    pos = [getattr(model, f"{m}_pos") for m in MODALS]
    intra = []
    for i, m in enumerate(MODALS):
        own5 = encs[i][4].with_mask(mask, i) if idt else encs[i][4]
        tokv = conv(getattr(model, f"{m}_encode_conv"), own5, shard, dg5, norm=False).t
        tok = GatherDFn.apply(tokv, shard)
        pH, pW = tok.shape[2], tok.shape[3]                     # token grid (Dg / 16, pH, pW): not necessarily a cube
        tok = tok.reshape(B, -1, model.token_dim)
        tok = model._transformer(getattr(model, f"{m}_transformer"), tok, pos[i])
        intra.append(tok * mask[:, i].view(B, 1, 1).to(tok.dtype))
    pos_cat = torch.cat(pos, 1)
    Dl5 = encs[0][4].t.shape[1]
    dgs = [Dg, Dg // 2, Dg // 4, Dg // 8, Dg // 16]
    df = model.decoder_fuse

    def rfm(mod, feats, dg):
        h = conv(mod.fusion_layer[0].conv, feats, shard, dg)
        h = conv(mod.fusion_layer[1].conv, h, shard, dg)
        return conv(mod.fusion_layer[2].conv, h, shard, dg)

    def fuse_pass(m_feat, m_tok):
        """one pass through the inter-modal transformer + decoder_fuse (reference mmformer.py:436-465 teacher, :469-531 students):
        m_feat (B,4) masks the encoder features, m_tok (B,4) or None the level-5 tokens.  -> logits, the four deep-supervision
        logits, the full-resolution feature map (de_x1)"""
        toks = intra if m_tok is None else [t * m_tok[:, i].view(B, 1, 1).to(t.dtype) for i, t in enumerate(intra)]
        tok = model._transformer(model.multimodal_transformer, torch.cat(toks, 1), pos_cat)
        volf = tok.to(dt).contiguous().view(B, dg5, pH, pW, model.token_dim * 4)
        mine = volf[:, shard.rank * Dl5:(shard.rank + 1) * Dl5].contiguous()
        x5 = conv(model.multimodal_decode_conv, Slab(mine, shard), shard, dg5, norm=False)
        xs = [[encs[i][l].with_mask(m_feat, i) if m_feat is not None else encs[i][l] for i in range(4)] for l in range(4)]
        f5 = rfm(df.RFM5, [x5], dgs[4])
        preds = {4: conv(df.seg_d4, f5, shard, dgs[4], norm=False, out_dtype=torch.float32).t}
        h = conv(df.d4_c1.conv, up(f5, 2, shard, dgs[4]), shard, dgs[3])
        f1 = None
        for lvl in (4, 3, 2, 1):
            r = rfm(getattr(df, f"RFM{lvl}"), xs[lvl - 1], dgs[lvl - 1])
            f = conv(getattr(df, f"d{lvl}_out").conv, conv(getattr(df, f"d{lvl}_c2").conv, [r, h], shard, dgs[lvl - 1]), shard, dgs[lvl - 1])
            if lvl > 1:
                preds[lvl - 1] = conv(getattr(df, f"seg_d{lvl - 1}"), f, shard, dgs[lvl - 1], norm=False, out_dtype=torch.float32).t
                h = conv(getattr(df, f"d{lvl - 1}_c1").conv, up(f, 2, shard, dgs[lvl - 1]), shard, dgs[lvl - 2])
            else:
                f1 = f
        return conv(df.seg_layer, f1, shard, Dg, norm=False, out_dtype=torch.float32).t, preds, f1.t

    logits, preds, f1_teacher = fuse_pass(mask if idt else None, None)
    pred_cl = ops.softmax_classes(logits, 1.0, "cl")

    tcl = ops.target_to_cl(target)
    # ---- per-modality decoders (shared weights, reference mmformer.py:66-114)
    ds = model.decoder_sep
    sep_loss = torch.zeros(B, 4, device=x.device)
    eye = torch.eye(4, device=x.device)
    maskf = mask.float()
    def sep_decoder(e):
        hh = conv(ds.d4_c1.conv, up(e[4], 2, shard, dgs[4]), shard, dgs[3])
        for lvl in (4, 3, 2, 1):
            hh = conv(getattr(ds, f"d{lvl}_out").conv, conv(getattr(ds, f"d{lvl}_c2").conv, [hh, e[lvl - 1]], shard, dgs[lvl - 1]),
                      shard, dgs[lvl - 1])
            if lvl > 1:
                hh = conv(getattr(ds, f"d{lvl - 1}_c1").conv, up(hh, 2, shard, dgs[lvl - 1]), shard, dgs[lvl - 2])
        return ops.softmax_classes(conv(ds.seg_layer, hh, shard, Dg, norm=False, out_dtype=torch.float32).t, 1.0, "cl")

    if stk is not None:                                       # shared weights: ONE pass over the stack (row i B + b = modality i of sample b)
        sep_mask = mask.t().reshape(4 * B, 1).contiguous() if idt else None
        sps = ops.unstack(sep_decoder([f.with_mask(sep_mask, 0) if idt else f for f in stk]), 4)
    else:
        sps = [sep_decoder([encs[i][l].with_mask(mask, i) if idt else encs[i][l] for l in range(5)]) for i in range(4)]
    for i in range(4):
        gate = (maskf * eye[i]) if idt else eye[i].expand(B, 4)
        sep_loss = sep_loss + gate * seg_pair(sps[i], tcl, shard).sum(1, keepdim=True)
    # ---- deep supervision (mmformer.py:540-544): softmax -> x 2^(j+1) -> WCE + Dice, weights 1/2, 1/4, 1/8, 1/16
    prm_loss = torch.zeros(B, 1, device=x.device)
    for j in range(4):
        pj = Slab(ops.softmax_classes(preds[j + 1], 1.0, "cl"), shard)
        prm_loss = prm_loss + 0.5 ** (j + 1) * seg_pair(up(pj, 2 ** (j + 1), shard, dgs[j + 1]).t, tcl, shard).sum(1, keepdim=True)
    fuse = seg_pair(pred_cl, tcl, shard).sum()
    mk = maskf if idt else torch.ones_like(maskf)
    pred = ops.to_ncdhw(pred_cl.detach())
    if not getattr(model, "use_passion", False):
        loss = fuse + (sep_loss * mk).sum() + prm_loss.sum()
        return loss, pred, prm_loss, sep_loss

    # ---- PASSION: the four uni-modal students (one pass each through the fusion path) against the detached teacher
    from .train import relative_preference
    inv_t = 1.0 / temp
    onehot = torch.eye(4, dtype=torch.bool, device=x.device)
    pt_full = ops.softmax_classes(logits.detach(), inv_t, "cl")
    pt_deep = [up(Slab(ops.softmax_classes(preds[j + 1].detach(), inv_t, "cl"), shard), 2 ** (j + 1), shard, dgs[j + 1]).t for j in range(4)]
    kl_loss, proto_loss, dist = (torch.zeros(B, 4, device=x.device) for _ in range(3))
    for i in range(4):
        m_i = onehot[i].expand(B, 4)
        m_feat = (mask & m_i) if idt else m_i                         # idt: absent-by-data modalities stay zero as well
        m_tok = onehot[2 if i == 3 else i].expand(B, 4)               # the T2 student masks its tokens with T1's one-hot (:522)
        lo_i, preds_i, f1_i = fuse_pass(m_feat, m_tok)
        kl = sharded_kl(ops.softmax_classes(lo_i, inv_t, "cl"), pt_full, temp, shard)
        for j in range(4):
            ps = up(Slab(ops.softmax_classes(preds_i[j + 1], inv_t, "cl"), shard), 2 ** (j + 1), shard, dgs[j + 1]).t
            kl = kl + 0.5 ** (j + 1) * sharded_kl(ps, pt_deep[j], temp, shard)
        pl, dd = sharded_proto(f1_i, f1_teacher, tcl, shard)
        gate = (maskf * eye[i]) if idt else eye[i].expand(B, 4)
        # a sample whose only present modality is i: student == teacher, the reference gets exact zeros (see models/mmformer.py)
        only = (mask & (mask.sum(1, keepdim=True) == 1))[:, i:i + 1]
        differs = torch.ones(B, 1, device=x.device) if i == 3 else (~only).float()
        kl_loss = kl_loss + gate * kl * differs
        proto_loss = proto_loss + gate * pl * differs
        dist = dist + gate * dd * differs
    beta = state.imb_beta if state is not None else torch.ones(4, device=x.device)
    w = state.modal_weight if (state is not None and idt) else torch.ones(4, device=x.device)
    red = (lambda t: (t * mk).sum(0))
    rp_mask = relative_preference(dist, mk, idt)
    loss = fuse + (rp_mask * beta * w * red(sep_loss)).sum() + prm_loss.sum() + 0.5 * (beta * w * red(kl_loss)).sum() \
        + 0.1 * (rp_mask * w * red(proto_loss)).sum()
    return loss, pred, prm_loss, sep_loss, kl_loss, proto_loss, dist


# ------------------------------------------------------------------------------------------------------------------
# the sharded training step, eager and as ONE HIP graph
# ------------------------------------------------------------------------------------------------------------------
def sharded_train_step(model, optimizer, shard, x, mask, target, Dg, temp=4.0, state=None, reducer=None):
    """one iteration on this rank's slab: sharded forward + loss, backward (halo gradients, InstanceNorm-backward sums and gathered
    tokens exchanged inside it), SUM of the weight gradients over the ranks, AdamW.  -> loss (detached; equal on every rank)"""
    host = L.host() if x.is_cuda else None
    direct = host is not None and (reducer is None or not reducer.active)      # as train.train_step: no AccumulateGrad nodes in one process
    model.is_training = True
    if direct:
        host.set_direct_grads(True)
        ops._DIRECT[0] = True
    try:
        loss = sharded_step_loss(model, shard, x, mask, target, Dg, temp, state)[0]
        optimizer.zero_grad()
        loss.backward()
    finally:
        if direct:
            host.set_direct_grads(False)
            ops._DIRECT[0] = False
    if reducer is not None:
        reducer.all_reduce_grads()
    optimizer.step()
    return loss.detach()


def GraphedDepthStep(model, optimizer, shard, Dg, temp=4.0, state=None, reducer=None, warm=2):
    """The sharded step recorded ONCE as a HIP graph and replayed (graph.GraphedStep with this step as its body): the eager form
    needs ~84 ms of host time per 240x240x160 volume -- per conv a halo pack / exchange / unpack, per InstanceNorm tensor a statistics
    all-reduce, a two-phase backward around the all-reduce of its sums, all issued from Python -- whatever the slab size, so it could
    not scale with the number of GPUs.  Recorded, the host's share is one hipGraphLaunch; the RCCL calls (batch_isend_irecv of the
    halo planes, all_reduce of the sums, all_gather of the level-5 tokens, the gradient buckets) are nodes of the same graph: torch's
    ProcessGroupNCCL runs them on its own stream behind events, which stream capture follows like any other fork."""
    from .graph import GraphedStep

    if os.environ.get("PASSION_DEPTH_HALO") != "p2p":
        shard.halo_by_gather = True          # from the first (eager) call on: the recorded step and its warm-up issue the same collectives

    def body(x, mask, target):
        return sharded_train_step(model, optimizer, shard, x, mask, target, Dg, temp, state, reducer)
    return GraphedStep(model, optimizer, temp, state, warm=warm, reducer=reducer, step_fn=body)
