"""torch.autograd.Function wrappers over the C ABI of libpassion_hip.so.

Internal activation layout is channels-last 5-D: (B, D, H, W, C) contiguous, f32 or bf16.
torch is used for memory, streams and autograd bookkeeping only -- every op below runs a
hand-written HIP kernel, and raises if the library or a GPU tensor is missing.
"""
from ctypes import byref, c_void_p

import os

import torch
import torch.nn.functional as F

from . import _lib as L


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _p_off(t, off_elems):
    return c_void_p(t.data_ptr() + off_elems * t.element_size())


# ------------------------------------------------------------------------------------------------
# InstanceNorm statistics
# ------------------------------------------------------------------------------------------------
def instnorm_stats(x, eps=1e-5, out=None):
    """(mean, rstd) per (b, c) of a channels-last (B,D,H,W,C) tensor -> f32 (B, C, 2).
    Statistics half of nn.InstanceNorm3d (reference blocks.py:18); not differentiated directly:
    the conv block's backward carries the InstanceNorm gradient analytically."""
    L.require_gpu(x)
    h = L.host()
    if h is not None:
        return h.instnorm_stats(x, eps, out)
    assert x.dim() == 5 and x.is_contiguous()
    B, D, H, W, C = x.shape
    V = D * H * W
    lib = L.load()
    scratch = torch.empty(lib.passion_instnorm_scratch_doubles(B, V, C), dtype=torch.float64, device=x.device)
    stats = out if out is not None else torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
    assert stats.shape == (B, C, 2) and stats.is_contiguous()
    rc = lib.passion_instnorm_stats(_p(x), L.dtype_code(x.dtype), B, V, C, V * C, eps,
                                         _p(scratch), _p(stats), L.stream_ptr())
    L.check(rc, "passion_instnorm_stats")
    return stats


# ------------------------------------------------------------------------------------------------
# conv block
# ------------------------------------------------------------------------------------------------
class SegIn:
    """One tensor of the virtual channel concatenation feeding a conv.

    x        : channels-last (B,D,H,W,C) tensor -- or any tensor when C/offset/bstride are given
               explicitly (used for the raw NCDHW input, one modality = one C=1 segment).
    stats    : (B,C,2) from instnorm_stats -> loader applies InstanceNorm + LeakyReLU; None -> raw.
    mask     : (B,K) bool tensor of present modalities and the column to read; absent => zeros.
    bmod     : > 0 -> the conv runs over a virtual batch of P*bmod samples (P masked passes over the
               same bmod physical samples: teacher + uni-modal students); sample b reads x[b % bmod],
               stats[b % bmod] and mask[b].  Its gradient is the sum over the P passes.
    """

    def __init__(self, x, stats=None, mask=None, mask_col=0, C=None, offset=0, bstride=None, bmod=0):
        self.x, self.stats, self.mask, self.mask_col, self.bmod = x, stats, mask, mask_col, bmod
        self.C = x.shape[-1] if C is None else C
        self.offset = offset
        self.bstride = (x[0].numel() if bstride is None else bstride)


class _ConvCfg:
    __slots__ = ("segs", "D", "H", "W", "ksize", "stride", "pad_mode", "slope", "out_dtype", "B", "out")


_PACK_REGISTRY = {}     # id(cache dict) -> (weakref to the weight parameter, cache dict)
_PACK_TABLE = {}        # the pointer tables of the last multi-weight re-pack


_PACK_MANAGED = [False]      # train.train_step re-packs right behind the optimizer (below): the call at the top of Model.forward has nothing to do


_PACK_F32 = [True]      # the f32 operand images are part of a weight pack (False: a bf16 model -- set_pack_f32)
_FORCE_F32 = [os.environ.get("PASSION_FORCE_F32_MFMA", "0") not in ("", "0")]      # bf16 tensors on the f32 kernels (library option force_f32_mfma)
PACK_BF16_ONLY = os.environ.get("PASSION_PACK_BF16_ONLY", "1") != "0"
_STICKY_F32 = [False]   # this process has run a bf16 conv that the library puts on its f32-image kernel (a concat with a segment whose channel
                        # count is no multiple of 8: m2ftrans): the f32 images stay part of every pack from then on


def _needs_f32_images(segs):
    """conv_bf16.hip bf16_eligible(): a multi-segment bf16 conv with a ragged segment runs on the general kernel, which reads the f32 images"""
    return segs[0].x.dtype != torch.bfloat16 or (len(segs) > 1 and any(s.C % 8 for s in segs))


def _pkey(w):
    """what a packed image was made from: the parameter's storage and version, and which images the pack wrote"""
    return (w.data_ptr(), w._version, _PACK_F32[0])


def set_pack_f32(on):
    """A bf16 model never reads the two f32 operand images of a weight pack (two thirds of the bytes a re-pack writes): its forward
    says so (models/*.py) and the library stops writing them (option pack_f32; its f32 kernels then refuse to run instead of reading
    stale images).  The image set is part of every pack's cache key: an f32 model in the same process re-packs what it needs."""
    on = bool(on) or _FORCE_F32[0] or _STICKY_F32[0] or not PACK_BF16_ONLY
    if on != _PACK_F32[0]:
        _PACK_F32[0] = on
        set_option("pack_f32", 1 if on else 0)


def repack_stale(only=None, exclude=None):
    """Re-pack, in ONE launch, every registered conv weight whose version moved since it was packed (i.e. all of
    them after an optimizer step) into its existing buffer.  Called at the top of Model.forward: the per-conv
    pack_weight() calls below then all hit their caches.
    only / exclude: sets of parameter data_ptr()s -- train.train_step packs the decoder / transformer weights right behind their
    early AdamW pass on the optimizer's stream (under the encoder backward) and the rest behind the closing step, so that the
    next forward does not start with a 0.14 ms pass over all weights on the critical stream."""
    capturing = torch.cuda.is_current_stream_capturing()   # a graph replays after optimizer steps: record all of them
    if capturing and _PACK_MANAGED[0] and only is None and exclude is None:
        return 0                                 # the recording holds the two packs behind the optimizer passes instead
    stale = []
    for key in list(_PACK_REGISTRY):
        ref, cache = _PACK_REGISTRY[key]
        w = ref()
        if w is None:
            del _PACK_REGISTRY[key]
            continue
        if only is not None and w.data_ptr() not in only:
            continue
        if exclude is not None and w.data_ptr() in exclude:
            continue
        if (capturing or cache.get("key") != _pkey(w)) and w.is_cuda and w.dtype == torch.float32 \
                and w.is_contiguous():
            stale.append((w, cache))
    if not stale:
        return 0
    n = len(stale)
    from ctypes import c_int as _ci
    sig = tuple((w.data_ptr(), cache["packed"].data_ptr()) for w, cache in stale)
    tabs = _PACK_TABLE.setdefault("tabs", {})
    tab = tabs.get(sig)
    if tab is None:                          # the same few hundred pointers every step (two sets with the early optimizer pass): build the tables once
        if len(tabs) > 8:
            tabs.clear()
        ptrs, outs = (c_void_p * n)(), (c_void_p * n)()
        cin, cout, ks = (_ci * n)(), (_ci * n)(), (_ci * n)()
        for i, (w, cache) in enumerate(stale):
            ptrs[i], outs[i] = w.data_ptr(), cache["packed"].data_ptr()
            cout[i], cin[i], ks[i] = w.shape[0], w.shape[1], w.shape[2]
        tab = tabs[sig] = (sig, ptrs, cin, cout, ks, outs)
    _, ptrs, cin, cout, ks, outs = tab
    L.check(L.load().passion_conv_pack_weights(n, ptrs, cin, cout, ks, outs, L.stream_ptr()), "passion_conv_pack_weights")
    for w, cache in stale:
        cache["key"] = _pkey(w)
    return n


def mark_packed_fresh():
    """after a graph replay that re-packed every registered image from the weights it had just written (graph.GraphedStep): the
    images match the parameters' CURRENT versions -- note that, so that the next repack_stale() only launches for weights somebody
    changed from outside the recording"""
    for key in list(_PACK_REGISTRY):
        ref, cache = _PACK_REGISTRY[key]
        w = ref()
        if w is None:
            del _PACK_REGISTRY[key]
        elif "packed" in cache:
            cache["key"] = _pkey(w)


# ------------------------------------------------------------------------------------------------
# bf16 images of the token-GEMM (nn.Linear) parameters
# ------------------------------------------------------------------------------------------------
_CAST_REGISTRY = {}     # id(param) -> [weakref to the f32 parameter, bf16 image or None, (data_ptr, version) of the image]


class _CastGradFn(torch.autograd.Function):
    """the cached bf16 image of an f32 parameter as a differentiable function of it (gradient: cast back to f32)"""

    @staticmethod
    def forward(ctx, param, image):
        return image.detach()

    @staticmethod
    def backward(ctx, g):
        return g.float(), None


def recast_stale(only=None, exclude=None):
    """Refresh, in one multi-tensor launch, the bf16 image of every registered Linear parameter the optimizer touched.
    Under autocast every F.linear casts its f32 weight and bias again on every call (40 tiny launches per step for the five
    transformers); a bf16 operand passes through autocast untouched.
    only / exclude: sets of parameter data_ptr()s, as repack_stale -- train.train_step casts the transformers' parameters right behind
    their early AdamW pass on the optimizer's stream (under the encoder backward: 157 MB that the next forward used to start with on
    the critical stream, 57 us per step) and whatever is left behind the closing step."""
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and _PACK_MANAGED[0] and only is None and exclude is None:
        return 0                                 # the recording holds the casts behind the optimizer passes instead
    src, dst = [], []
    for key in list(_CAST_REGISTRY):
        ent = _CAST_REGISTRY[key]
        p = ent[0]()
        if p is None:
            del _CAST_REGISTRY[key]
            continue
        if not p.is_cuda:
            continue
        if only is not None and p.data_ptr() not in only:
            continue
        if exclude is not None and p.data_ptr() in exclude:
            continue
        if ent[1] is None or ent[1].device != p.device:
            ent[1], ent[2] = torch.empty_like(p, dtype=torch.bfloat16), None
        if capturing or ent[2] != (p.data_ptr(), p._version):
            src.append(p.detach())
            dst.append(ent[1])
            ent[2] = (p.data_ptr(), p._version)
    if src:
        torch._foreach_copy_(dst, src)
    return len(src)


def mark_cast_fresh():
    """after a graph replay that re-cast every registered image from the parameters it had just written (graph.GraphedStep): the images
    match the parameters' current versions (mark_packed_fresh, for the token GEMMs' operands)"""
    for key in list(_CAST_REGISTRY):
        ent = _CAST_REGISTRY[key]
        p = ent[0]()
        if p is None:
            del _CAST_REGISTRY[key]
        elif ent[1] is not None:
            ent[2] = (p.data_ptr(), p._version)


def bf16_image(p):
    """the (detached) bf16 image of the f32 parameter `p`, registered on first use and refreshed by recast_stale"""
    ent = _CAST_REGISTRY.get(id(p))
    if ent is None or ent[0]() is not p:
        import weakref
        ent = _CAST_REGISTRY[id(p)] = [weakref.ref(p), None, None]
    if ent[1] is None or ent[2] != (p.data_ptr(), p._version):
        recast_stale()
    return ent[1]


def bf16_param(p):
    """bf16 image of the f32 parameter `p` (registered on first use; refreshed by recast_stale at the top of forward)"""
    ent = _CAST_REGISTRY.get(id(p))
    if ent is None or ent[0]() is not p:
        import weakref
        ent = _CAST_REGISTRY[id(p)] = [weakref.ref(p), None, None]
    if ent[1] is None or ent[2] != (p.data_ptr(), p._version):
        recast_stale()
    return _CastGradFn.apply(p, ent[1])


# Direct-gradient mode (train.train_step, single process; csrc/host_ext.cpp has the conv nodes' half): the Functions below are
# then given their parameters inside an opaque box instead of as tensor arguments and write `.grad` themselves (assign when it is
# None, add otherwise -- what AccumulateGrad does), so autograd builds no AccumulateGrad node per parameter: a few hundred nodes
# less per step for an engine whose per-node cost is host time (m2ftrans, whose step is bound by the host, has ~700 of them).
_DIRECT = [False]


class _Box:
    __slots__ = ("params",)

    def __init__(self, params):
        self.params = tuple(params)


def _give(p, g):
    if p is None or g is None or not p.requires_grad:
        return
    with torch.no_grad():
        if p.grad is None:
            p.grad = g
        else:
            p.grad.add_(g)


def _direct(x, params):
    """the mode is on, the node stays in the graph through x, and every parameter is a LEAF (a padded / sliced view of a parameter
    has to send its gradient on through autograd)"""
    return _DIRECT[0] and torch.is_grad_enabled() and x.requires_grad and all(p is None or p.is_leaf for p in params)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b for the token GEMMs (reference mmformer.py nn.Linear): the GEMMs stay on torch (hipBLASLt),
    the bias gradient is the library's fixed-order column sum (passion_colsum) instead of torch's `grad.sum(0)`, whose reduce
    kernel does not repeat -- and drops contributions -- while the encoders' side streams keep the GPU busy
    (tools/linear_bias_determinism.py).  weight / bias are the f32 parameters (they receive f32 gradients: the weight
    gradient GEMM writes f32 directly where torch can, no bf16 rounding + cast pass); w / b the operands in the compute
    dtype (their kept bf16 images, or the parameters themselves)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w, b, box=None):
        ctx.save_for_backward(x, w)
        ctx.box = box                         # direct-gradient mode: (weight, bias) in here, the tensor slots hold None
        ctx.wshape = (box.params[0] if box is not None else weight).shape
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        box = ctx.box
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        dx = torch.mm(g2, w).view(x.shape) if ctx.needs_input_grad[0] else None
        need_w = box.params[0].requires_grad if box is not None else ctx.needs_input_grad[1]
        need_b = (box.params[1] is not None and box.params[1].requires_grad) if box is not None else ctx.needs_input_grad[2]
        dw = _mm_f32(g2.t(), x2).view(ctx.wshape) if need_w else None
        db = colsum(g2) if need_b else None
        if box is not None:
            _give(box.params[0], dw)
            _give(box.params[1], db)
            dw = db = None
        return dx, dw, db, None, None, None


_MM_F32 = [None]      # torch.mm(bf16, bf16, out_dtype=f32) exists on this build?


def _mm_f32(a, b):
    """a @ b with an f32 result (the weight gradient of a token GEMM)"""
    if a.dtype == torch.float32:
        return torch.mm(a, b)
    if _MM_F32[0] is None:
        try:
            torch.mm(a[:, :8].contiguous()[:8], b[:8], out_dtype=torch.float32)
            _MM_F32[0] = True
        except Exception:
            _MM_F32[0] = False
    if _MM_F32[0]:
        return torch.mm(a, b, out_dtype=torch.float32)
    return torch.mm(a, b).float()


def colsum(g2):
    """f32 column sums of a contiguous [R][C] f32 / bf16 matrix, fixed summation order"""
    L.require_gpu(g2)
    R, C = g2.shape
    lib = L.load()
    out = torch.empty(C, dtype=torch.float32, device=g2.device)
    work = torch.empty(lib.passion_colsum_work_floats(C), dtype=torch.float32, device=g2.device)
    L.check(lib.passion_colsum(_p(g2), L.dtype_code(g2.dtype), R, C, _p(out), _p(work), L.stream_ptr()), "passion_colsum")
    return out


def linear(x, weight, bias, bf16):
    """nn.Linear forward for a token GEMM: bf16 -> the kept bf16 operand images (refreshed by recast_stale), else the f32
    parameters"""
    if bf16:
        w, b = bf16_image(weight), (bf16_image(bias) if bias is not None else None)
    else:
        w, b = weight.detach(), (bias.detach() if bias is not None else None)
    if w.dim() > 2:              # the (Cout, Cin, 1, 1, 1) weight of a 1x1x1 conv run as a token GEMM
        w = w.view(w.shape[0], -1)
    if _direct(x, (weight, bias)):
        return LinearFn.apply(x, None, None, w, b, _Box((weight, bias)))
    return LinearFn.apply(x, weight, bias, w, b)


def colsum_batched(g3):
    """f32 column sums of each matrix of a contiguous [G][R][C] tensor -> [G][C], one launch, fixed summation order"""
    L.require_gpu(g3)
    G, R, C = g3.shape
    lib = L.load()
    out = torch.empty(G, C, dtype=torch.float32, device=g3.device)
    work = torch.empty(G * lib.passion_colsum_work_floats(C), dtype=torch.float32, device=g3.device)
    L.check(lib.passion_colsum_batched(_p(g3), L.dtype_code(g3.dtype), G, R, C, _p(out), _p(work), L.stream_ptr()), "passion_colsum_batched")
    return out


_GROUP_IMAGES = {}     # ids of the G parameters -> [G, *shape] bf16 buffer whose slices are the parameters' registered images


def bf16_group_image(params):
    """the bf16 images of G same-shaped f32 parameters back to back in one (G, *shape) tensor: each parameter's registered image
    (bf16_image / recast_stale) IS its slice, so one multi-tensor refresh keeps the stack current"""
    key = tuple(id(p) for p in params)
    ent = _GROUP_IMAGES.get(key)
    p0 = params[0]
    if ent is None or ent[0].device != p0.device or any(r() is not p for r, p in zip(ent[1], params)):
        import weakref
        buf = torch.empty((len(params),) + tuple(p0.shape), dtype=torch.bfloat16, device=p0.device)
        for g, p in enumerate(params):
            _CAST_REGISTRY[id(p)] = [weakref.ref(p), buf[g], None]
        ent = _GROUP_IMAGES[key] = (buf, [weakref.ref(p) for p in params])
    if any(_CAST_REGISTRY[id(p)][2] != (p.data_ptr(), p._version) for p in params):
        recast_stale()
    return ent[0]


class GroupedLinearFn(torch.autograd.Function):
    """G token GEMMs with G weight sets in one batched GEMM: y[g] = x[g] W[g]^T + b[g] (the four intra-modal transformers of
    mmformer share every shape, reference mmformer.py:390-393).  x: (G, R, Cin); w: (G, Cout, Cin) / b: (G, Cout) operands in the
    compute dtype; the 2G parameters are the autograd inputs that receive f32 gradients (bias: fixed-order column sums)."""

    @staticmethod
    def forward(ctx, x, w, b, n, *params):
        ctx.save_for_backward(x, w)
        ctx.box = None
        if len(params) == 1 and isinstance(params[0], _Box):      # direct-gradient mode
            ctx.box = params[0]
            params = ctx.box.params
        ctx.n, ctx.has_bias, ctx.wshape = n, b is not None, params[0].shape
        if b is None:
            return torch.bmm(x, w.transpose(1, 2))
        return torch.baddbmm(b.unsqueeze(1), x, w.transpose(1, 2))

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        n = ctx.n
        g = g.contiguous()
        dx = torch.bmm(g, w) if ctx.needs_input_grad[0] else None
        if g.dtype == torch.float32:
            dw = torch.bmm(g.transpose(1, 2), x)
        elif _mm_f32_ok(g):
            dw = torch.bmm(g.transpose(1, 2), x, out_dtype=torch.float32)
        else:
            dw = torch.bmm(g.transpose(1, 2), x).float()
        grads = [dw[i].view(ctx.wshape) for i in range(n)]
        if ctx.has_bias:
            db = colsum_batched(g)
            grads += [db[i] for i in range(n)]
        if ctx.box is not None:
            for p_, g_ in zip(ctx.box.params, grads):
                _give(p_, g_)
            return dx, None, None, None, None
        return (dx, None, None, None, *grads)


def _mm_f32_ok(t):
    if _MM_F32[0] is None:
        _mm_f32(t.reshape(-1, t.shape[-1])[:8].t().contiguous(), t.reshape(-1, t.shape[-1])[:8].contiguous())
    return bool(_MM_F32[0])


def linear_grouped(x, weights, biases, bf16):
    """x: (G, R, Cin) -> (G, R, Cout) with weight set g for matrix g; weights / biases: lists of G parameters (biases may be None)"""
    n = len(weights)
    if bf16:
        w = bf16_group_image(weights)
        b = bf16_group_image(biases) if biases is not None else None
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
    else:
        w = torch.stack([p.detach() for p in weights])
        b = torch.stack([p.detach() for p in biases]) if biases is not None else None
    if w.dim() > 3:              # (G, Cout, Cin, 1, 1, 1) weights of 1x1x1 convs run as token GEMMs
        w = w.view(n, w.shape[1], -1)
    params = list(weights) + (list(biases) if biases is not None else [])
    if _direct(x, params):
        return GroupedLinearFn.apply(x, w, b, n, _Box(params))
    return GroupedLinearFn.apply(x, w, b, n, *params)


def pack_weight(weight, cache=None):
    """f32 (Cout,Cin,k,k,k) -> packed MFMA operand images (fwd + dgrad); cached per weight version."""
    key = _pkey(weight)
    if cache is not None and cache.get("key") == key:
        return cache["packed"]
    L.require_gpu(weight)
    Cout, Cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
    lib = L.load()
    n = lib.passion_conv_pack_floats(Cin, Cout, k)
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    L.check(lib.passion_conv_pack_weight(_p(w), Cin, Cout, k, _p(packed), L.stream_ptr()),
            "passion_conv_pack_weight")
    if cache is not None:
        cache["key"], cache["packed"] = key, packed
        if weight.is_leaf and id(cache) not in _PACK_REGISTRY:
            import weakref
            _PACK_REGISTRY[id(cache)] = (weakref.ref(weight), cache)
    return packed


def _fill_desc(cfg, xs, dxs=None):
    d = L.ConvDesc()
    d.x_dtype = L.dtype_code(xs[0].dtype) if xs[0].dtype != torch.float64 else L.F32
    d.y_dtype = L.dtype_code(cfg.out_dtype)
    d.B, d.D, d.H, d.W = cfg.B, cfg.D, cfg.H, cfg.W
    d.ksize, d.stride, d.pad_mode = cfg.ksize, cfg.stride, cfg.pad_mode
    d.nseg = len(cfg.segs)
    d.lrelu_slope = cfg.slope
    for i, (s, x) in enumerate(zip(cfg.segs, xs)):
        sg = d.seg[i]
        sg.x = x.data_ptr() + s.offset * x.element_size()
        sg.stats = s.stats.data_ptr() if s.stats is not None else None
        sg.mask = s.mask.data_ptr() if s.mask is not None else None
        sg.mask_col = s.mask_col
        sg.mask_ld = s.mask.shape[1] if s.mask is not None else 0
        sg.C = s.C
        sg.bstride = s.bstride
        sg.bmod = s.bmod
        sg.dx = dxs[i].data_ptr() if (dxs is not None and dxs[i] is not None) else None
    return d


class ConvBlockFn(torch.autograd.Function):
    """y = Conv3d(LeakyReLU(InstanceNorm(masked concat(xs)))) + bias (+ residual)
    -- general_conv3d_prenorm (reference blocks.py:300-316) with the norm/activation/mask/pad/concat
    folded into the conv's loader; with stats=None it is a bare nn.Conv3d."""

    @staticmethod
    def forward(ctx, cfg, weight, bias, residual, packed, *xs):
        L.require_gpu(weight, *xs)
        lib = L.load()
        Cout = weight.shape[0]
        pad = cfg.ksize // 2
        Do = (cfg.D + 2 * pad - cfg.ksize) // cfg.stride + 1
        Ho = (cfg.H + 2 * pad - cfg.ksize) // cfg.stride + 1
        Wo = (cfg.W + 2 * pad - cfg.ksize) // cfg.stride + 1
        y, cfg.out = cfg.out, None
        if y is None:
            y = torch.empty(cfg.B, Do, Ho, Wo, Cout, dtype=cfg.out_dtype, device=weight.device)
        assert y.shape == (cfg.B, Do, Ho, Wo, Cout) and y.dtype == cfg.out_dtype and y.is_contiguous()
        d = _fill_desc(cfg, xs)
        d.Cout = Cout
        b32 = bias.detach().float() if bias is not None else None
        if residual is not None:
            assert residual.shape == y.shape and residual.dtype == y.dtype and residual.is_contiguous()
        L.check(lib.passion_conv3d_fwd(byref(d), _p(packed), _p(b32), _p(residual), _p(y),
                                       L.stream_ptr()), "passion_conv3d_fwd")
        ctx.cfg, ctx.packed, ctx.has_bias, ctx.has_res = cfg, packed, bias is not None, residual is not None
        ctx.wshape = weight.shape
        ctx.save_for_backward(*xs)
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg, xs = ctx.cfg, ctx.saved_tensors
        lib = L.load()
        dy = dy.contiguous()
        need_w = ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dev = dy.device
        dW = db = None
        if need_w and need_b:      # one buffer [dW | db]: the library folds both partial slabs in one launch
            nW = ctx.wshape.numel()
            buf = torch.empty(nW + ctx.wshape[0], dtype=torch.float32, device=dev)
            dW, db = buf[:nW].view(ctx.wshape), buf[nW:]
        elif need_w:
            dW = torch.empty(ctx.wshape, dtype=torch.float32, device=dev)
        elif need_b:
            db = torch.empty(ctx.wshape[0], dtype=torch.float32, device=dev)
        dxs = []
        for i, (s, x) in enumerate(zip(cfg.segs, xs)):
            if ctx.needs_input_grad[5 + i]:
                assert s.offset == 0, "gradient wrt an offset view is not supported"
                dxs.append(torch.empty_like(x))     # bmod segments: the library sums the passes into the physical samples
            else:
                dxs.append(None)
        d = _fill_desc(cfg, xs, dxs)
        d.Cout = ctx.wshape[0]
        sums = torch.empty(lib.passion_conv_bwd_scratch_floats(byref(d)), dtype=torch.float32, device=dev)
        L.check(lib.passion_conv3d_bwd(byref(d), _p(ctx.packed), _p(dy), _p(dW), _p(db), _p(sums),
                                       L.stream_ptr()), "passion_conv3d_bwd")
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return (None, dW, db, dres, None, *dxs)


def conv_block(segs, weight, bias, D, H, W, ksize=3, stride=1, pad_mode=L.PAD_REFLECT,
               residual=None, out_dtype=None, slope=0.2, pack_cache=None, out=None, want_stats=False, stats_out=None, eps=1e-5):
    """`out`: optional preallocated output (a non-view alias into a larger buffer, see alias_slice).
    want_stats: return (y, InstanceNorm statistics of y) -- the statistics the NEXT pre-norm conv applies in its loader, left
    by this conv's own epilogue where the kernel can (passion_conv3d_fwd_stats); `stats_out`: preallocated (B,Cout,2) slot."""
    cfg = _ConvCfg()
    cfg.segs, cfg.D, cfg.H, cfg.W = segs, D, H, W
    cfg.ksize, cfg.stride, cfg.pad_mode, cfg.slope = ksize, stride, pad_mode, slope
    vb = [s.mask.shape[0] if s.bmod else s.x.shape[0] for s in segs]
    assert min(vb) == max(vb), "segments disagree on the (virtual) batch size"
    cfg.B, cfg.out = vb[0], out
    x0 = segs[0].x
    cfg.out_dtype = out_dtype if out_dtype is not None else (x0.dtype if x0.dtype != torch.float64 else torch.float32)
    if not _PACK_F32[0] and _needs_f32_images(segs):
        _STICKY_F32[0] = _STICKY_F32[0] or x0.dtype == torch.bfloat16
        set_pack_f32(True)                       # (an f32 conv after a bf16 model in this process, or a ragged concat: all four images again)
    packed = pack_weight(weight, pack_cache)
    h = L.host()
    if h is not None:            # C++ autograd node (csrc/host_ext.cpp): same descriptor, same C-ABI calls
        L.require_gpu(weight, x0)
        seg_ints = []
        for s in segs:
            seg_ints += (s.C, s.offset, s.bstride, s.bmod, s.mask_col)
        r = h.conv_block(weight, bias, residual, [s.x for s in segs], [s.stats for s in segs], [s.mask for s in segs], seg_ints,
                         (cfg.B, D, H, W, ksize, stride, pad_mode, L.dtype_code(cfg.out_dtype)), slope, packed, out,
                         want_stats, stats_out, eps)
        return (r[0], r[1]) if want_stats else r[0]
    y = ConvBlockFn.apply(cfg, weight, bias, residual, packed, *[s.x for s in segs])
    return (y, instnorm_stats(y.detach(), eps, out=stats_out)) if want_stats else y


_GROUP_PACKS = {}     # ids of the G weights -> packed workspace of the group (the G images back to back)


def _group_pack(weights):
    """one workspace holding the packed images of the G weights at equal spacing; each weight's own pack cache points into it, so
    repack_stale() refreshes the group with everything else"""
    key = tuple(id(w) for w in weights)
    ent = _GROUP_PACKS.get(key)
    w0 = weights[0]
    if ent is not None and ent[0].device == w0.device and all(r() is w for r, w in zip(ent[1], weights)):
        if any(c.get("key") != _pkey(w) for c, w in zip(ent[2], weights)):
            repack_stale()
        return ent[0]
    import weakref
    lib = L.load()
    Cout, Cin, k = w0.shape[0], w0.shape[1], w0.shape[2]
    n = lib.passion_conv_pack_floats(Cin, Cout, k)
    n4 = (n + 3) // 4 * 4
    buf = torch.empty(len(weights) * n4, dtype=torch.float32, device=w0.device)
    caches = []
    for g, w in enumerate(weights):
        cache = {"packed": buf[g * n4:g * n4 + n]}
        wd = w.detach()
        L.check(lib.passion_conv_pack_weight(_p(wd), Cin, Cout, k, _p(cache["packed"]), L.stream_ptr()), "passion_conv_pack_weight")
        cache["key"] = _pkey(w)
        if w.is_leaf:
            _PACK_REGISTRY[id(cache)] = (weakref.ref(w), cache)
        caches.append(cache)
    _GROUP_PACKS[key] = (buf, [weakref.ref(w) for w in weights], caches)
    return buf


def conv_block_grouped(segs, weights, biases, D, H, W, ksize=3, stride=1, pad_mode=L.PAD_REFLECT, residual=None, out_dtype=None,
                       slope=0.2, out=None, want_stats=False, stats_out=None, eps=1e-5, pass_x=False):
    """conv_block for G weight sets over one stacked batch: sample b of the (B,...) segments uses weights[b // (B // G)] -- the
    four modality encoders of mmformer as one launch per layer (passion_conv3d_fwd_grouped / _bwd_grouped).
    pass_x: the single segment's tensor comes back as a further result (see below)."""
    h = L.host()
    assert h is not None, "grouped conv blocks need the C++ host extension"
    x0 = segs[0].x
    L.require_gpu(weights[0], x0)
    B = x0.shape[0]
    assert B % len(weights) == 0 and all(s.bmod == 0 for s in segs)
    odt = out_dtype if out_dtype is not None else (x0.dtype if x0.dtype != torch.float64 else torch.float32)
    if not _PACK_F32[0] and _needs_f32_images(segs):
        _STICKY_F32[0] = _STICKY_F32[0] or x0.dtype == torch.bfloat16
        set_pack_f32(True)
    packed = _group_pack(weights)
    seg_ints = []
    for s in segs:
        seg_ints += (s.C, s.offset, s.bstride, s.bmod, s.mask_col)
    r = h.conv_block_grouped(list(weights), list(biases) if biases is not None else [], residual, [s.x for s in segs],
                             [s.stats for s in segs], [s.mask for s in segs], seg_ints,
                             (B, D, H, W, ksize, stride, pad_mode, L.dtype_code(odt)), slope, packed, out, want_stats, stats_out, eps, int(pass_x))
    if pass_x:       # ... and the input again (pass_x = 1 or 2 times), as outputs of this node: give THOSE to whoever else reads x (a later
        #              conv's `residual`, the decoders) and the gradients of x meet inside this conv's backward (passion_seg_t.dx_add,
        #              dx_add2) instead of in autograd adds
        n = int(pass_x)
        return tuple(r[:2 if want_stats else 1]) + tuple(r[len(r) - n:])
    return (r[0], r[1]) if want_stats else r[0]


def alias_slice(buf, i):
    """buf[i] as a tensor that shares buf's storage but is NOT an autograd view of it, so that an op may
    return it as its own output (autograd forbids custom-Function outputs that are views of a base which is
    written elsewhere).  Used to let producers write straight into one slot of a stacked buffer."""
    sub = buf[i]
    return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(buf.untyped_storage(), sub.storage_offset(),
                                                                    sub.shape, sub.stride())


class StackFn(torch.autograd.Function):
    """torch.cat(parts, 0) where the parts already live back to back in `buf` (alias_slice): no copy forward,
    the gradient is handed out in chunks backward."""

    @staticmethod
    def forward(ctx, holder, *parts):
        buf = holder[0]
        n, b = buf.shape[0], buf.shape[1]
        assert len(parts) == n and all(p.data_ptr() == buf[i].data_ptr() for i, p in enumerate(parts))
        ctx.n = n
        return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(
            buf.untyped_storage(), buf.storage_offset(), (n * b,) + tuple(buf.shape[2:]), buf.stride()[1:])

    @staticmethod
    def backward(ctx, g):
        return (None, *g.contiguous().chunk(ctx.n, 0))


class UnstackFn(torch.autograd.Function):
    """The n equal batch slices of a stacked (n*B, ...) tensor as tensors that share its storage without being autograd views of
    it (the four modalities of a grouped encoder level, read as four segments by the fusion convs).  Backward: when the n
    gradients are adjacent slices of one buffer (the conv backward allocates them that way, csrc/host_ext.cpp) the buffer IS
    the gradient of the stack; otherwise they are concatenated."""

    @staticmethod
    def forward(ctx, t, n):
        assert t.is_contiguous() and t.shape[0] % n == 0
        B = t.shape[0] // n
        ctx.n, ctx.shape, ctx.dtype = n, t.shape, t.dtype
        ctx.set_materialize_grads(False)
        return tuple(torch.empty(0, dtype=t.dtype, device=t.device).set_(
            t.untyped_storage(), t.storage_offset() + i * B * t.stride(0), (B,) + tuple(t.shape[1:]), t.stride()) for i in range(n))

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        if all(g is None for g in gs):
            return None, None
        g0 = gs[0]
        if g0 is not None and all(g is not None and g.is_contiguous() and g.shape == g0.shape and g.dtype == g0.dtype for g in gs) and \
                all(gs[i].data_ptr() == g0.data_ptr() + i * g0.numel() * g0.element_size() for i in range(n)) and \
                g0.untyped_storage().nbytes() - g0.storage_offset() * g0.element_size() >= n * g0.numel() * g0.element_size():
            return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), g0.storage_offset(), ctx.shape), None
        B = ctx.shape[0] // n
        ref = next(g for g in gs if g is not None)
        parts = [g if g is not None else torch.zeros((B,) + tuple(ctx.shape[1:]), dtype=ref.dtype, device=ref.device) for g in gs]
        return torch.cat(parts, 0), None


def unstack(t, n):
    return UnstackFn.apply(t, n)


_MARKER_CB = [None]


class DeferredFn(torch.autograd.Function):
    """y = f(x) where f was already RUN, earlier, on a detached copy `xd` of x (`early` = f(xd), with its own small autograd graph):
    this node only attaches that graph to x.  The forward kernels of f then sit where they were launched (a side stream, under other
    work) while the node -- and with it the place of f's backward in the engine's order, newest node first -- is where this call is.
    Used for the deep-supervision losses (models/mmformer.py): their forward runs as soon as a level's logits exist instead of in a
    chain between the end of the forward and the start of the backward."""

    @staticmethod
    def forward(ctx, x, box):
        # box = [early, xd] as a plain list: handed over as tensors they would be differentiable INPUTS of this node and the engine
        # would walk the little graph itself (with undefined gradients) besides the call below
        ctx.box = box
        return box[0].detach()

    @staticmethod
    def backward(ctx, g):
        early, xd = ctx.box
        ctx.box = None
        (gx,) = torch.autograd.grad(early, xd, g)
        return gx, None


class BackwardMarkerFn(torch.autograd.Function):
    """identity whose backward node tells the training step how far the backward has come.  Model.forward puts it on the deepest
    encoder output: autograd runs nodes in reverse creation order among the ready ones, every decoder / transformer / loss node
    was created after every encoder node and none of them depends on an encoder node's backward, so when this node runs all
    of THEIR parameter gradients have been enqueued and only the encoders' backward is left (train.train_step uses that to start
    the optimizer on those parameters under the encoder backward)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        cb = _MARKER_CB[0]
        if cb is not None:
            cb()
        return g


def backward_marker(x):
    return BackwardMarkerFn.apply(x) if (x.requires_grad and _MARKER_CB[0] is not None) else x


def stack_slots(buf, parts):
    return StackFn.apply([buf], *parts)


# ------------------------------------------------------------------------------------------------
# trilinear upsampling
# ------------------------------------------------------------------------------------------------
class UpsampleFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=k, mode='trilinear', align_corners=True) on (B,D,H,W,C)."""

    @staticmethod
    def forward(ctx, x, k, want_stats=False, eps=1e-5):
        L.require_gpu(x)
        x = x.contiguous()
        B, D, H, W, C = x.shape
        y = torch.empty(B, D * k, H * k, W * k, C, dtype=x.dtype, device=x.device)
        ctx.k, ctx.shape = k, (B, D, H, W, C)
        if want_stats:       # the InstanceNorm statistics of y from the same pass (the next conv normalises y in its loader)
            st = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
            L.check(L.load().passion_upsample_fwd_stats(_p(x), _p(y), L.dtype_code(x.dtype), B, D, H, W, C, k, _p(st), eps,
                                                        L.stream_ptr()), "passion_upsample_fwd_stats")
            ctx.mark_non_differentiable(st)
            ctx.set_materialize_grads(False)
            return y, st
        L.check(L.load().passion_upsample_fwd(_p(x), _p(y), L.dtype_code(x.dtype), B, D, H, W, C, k,
                                              L.stream_ptr()), "passion_upsample_fwd")
        return y

    @staticmethod
    def backward(ctx, dy, _dst=None):
        if dy is None:
            return None, None, None, None
        B, D, H, W, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, D, H, W, C, dtype=dy.dtype, device=dy.device)
        lib = L.load()
        ns = lib.passion_upsample_bwd_scratch_floats(B, D, H, W, C, ctx.k)
        scratch = torch.empty(ns, dtype=torch.float32, device=dy.device) if ns else None
        L.check(lib.passion_upsample_bwd(_p(dy), _p(dx), L.dtype_code(dy.dtype), B, D, H, W, C,
                                         ctx.k, _p(scratch), L.stream_ptr()), "passion_upsample_bwd")
        return dx, None, None, None


def upsample(x, k):
    return x if k == 1 else UpsampleFn.apply(x, k)


def upsample_with_stats(x, k, eps=1e-5):
    """-> (upsampled tensor, its InstanceNorm statistics (B,C,2))"""
    return UpsampleFn.apply(x, k, True, eps)


# ------------------------------------------------------------------------------------------------
# class softmax, target layout
# ------------------------------------------------------------------------------------------------
def _strides(t, layout):
    """element strides (batch, voxel, class) of a 4-class f32 tensor in 'cl' or 'ncdhw' layout"""
    if layout == "cl":
        B, D, H, W, C = t.shape
        V = D * H * W
        return B, V, C, (V * C, C, 1)
    B, C, D, H, W = t.shape
    V = D * H * W
    return B, V, C, (C * V, 1, V)


class SoftmaxFn(torch.autograd.Function):
    """softmax(logits * inv_temp) over the class axis.  logits: f32 channels-last (B,D,H,W,4);
    output channels-last or NCDHW (B,4,D,H,W) -- the layout the reference Model returns."""

    @staticmethod
    def forward(ctx, logits, inv_temp, out_layout):
        L.require_gpu(logits)
        logits = logits.contiguous()
        assert logits.dtype == torch.float32
        B, D, H, W, C = logits.shape
        V = D * H * W
        if out_layout == "cl":
            probs = torch.empty_like(logits)
        else:
            probs = torch.empty(B, C, D, H, W, dtype=torch.float32, device=logits.device)
        _, _, _, ps = _strides(probs, out_layout)
        L.check(L.load().passion_softmax_fwd(_p(logits), V * C, C, 1, _p(probs), *ps, B, V, C,
                                             float(inv_temp), L.stream_ptr()), "passion_softmax_fwd")
        ctx.inv_temp, ctx.layout = float(inv_temp), out_layout
        ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        (probs,) = ctx.saved_tensors
        dprobs = dprobs.contiguous()
        B, V, C, ps = _strides(probs, ctx.layout)
        if ctx.layout == "cl":
            dl = torch.empty_like(probs)
        else:
            dl = torch.empty(B, probs.shape[2], probs.shape[3], probs.shape[4], C, dtype=torch.float32,
                             device=probs.device)
        L.check(L.load().passion_softmax_bwd(_p(probs), *ps, _p(dprobs), *ps, _p(dl), V * C, C, 1, B, V, C,
                                             ctx.inv_temp, L.stream_ptr()), "passion_softmax_bwd")
        return dl, None, None


def softmax_classes(logits, inv_temp=1.0, out_layout="cl"):
    return SoftmaxFn.apply(logits, inv_temp, out_layout)


_TCL_STEP = [0]      # train.train_step counts its calls here: a conversion made under stream capture is valid for THAT step's recording only


def target_to_cl(target):
    """one-hot target (B,4,D,H,W) f64/f32 (datasets_nii.py:151-153) -> f32 channels-last (B,D,H,W,4).
    The converted image is kept on the tensor OBJECT: Model.forward and the loss assembly of one step (train.assemble_loss ->
    criterions.seg_pair_bs) convert once, also inside a recording (round 5 converted twice per recorded step: the cache was bypassed
    while capturing).  A target that already IS channels-last in memory (one_hot(...).permute(0, 4, 1, 2, 3): what train.synthetic_batch
    and a loader that permutes instead of copying hand over) is read through its strides -- `.contiguous()` used to copy the 32-byte-
    per-voxel float64 volume to NCDHW first, only for the kernel to transpose it back."""
    L.require_gpu(target)
    capturing = torch.cuda.is_current_stream_capturing()
    cached = getattr(target, "_passion_cl", None)
    if cached is not None and cached[0] == target._version:
        if len(cached) == 2:                       # made by a loader (data/transforms.py): plain data, valid outside recordings
            if not capturing:
                return cached[1]
        elif cached[2] == (capturing, _TCL_STEP[0] if capturing else -1):
            return cached[1]
    B, C, D, H, W = target.shape
    t = target
    if t.dtype not in (torch.float32, torch.float64):
        t = t.float()
    out = torch.empty(B, D, H, W, C, dtype=torch.float32, device=t.device)
    if t.permute(0, 2, 3, 4, 1).is_contiguous() and not t.is_contiguous():
        out.copy_(t.permute(0, 2, 3, 4, 1))        # same element order in memory: one streaming conversion
    else:
        t = t.contiguous()
        L.check(L.load().passion_target_to_cl(_p(t), 1 if t.dtype == torch.float64 else 0, _p(out), B, D * H * W, C,
                                              L.stream_ptr()), "passion_target_to_cl")
    try:
        target._passion_cl = (target._version, out, (capturing, _TCL_STEP[0] if capturing else -1))
    except Exception:
        pass
    return out


def to_channels_last(x, dtype=None):
    """(B,C,D,H,W) -> (B,D,H,W,C) in `dtype` through the strided-copy kernel (no autograd)."""
    L.require_gpu(x)
    x = x.contiguous()
    B, C, D, H, W = x.shape
    V = D * H * W
    dtype = dtype or x.dtype
    out = torch.empty(B, D, H, W, C, dtype=dtype, device=x.device)
    L.check(L.load().passion_copy_strided(_p(x), L.dtype_code(x.dtype), C * V, 1, V, _p(out), L.dtype_code(dtype),
                                          V * C, C, 1, B, V, C, L.stream_ptr()), "passion_copy_strided")
    return out


def to_ncdhw(x, dtype=None):
    L.require_gpu(x)
    x = x.contiguous()
    B, D, H, W, C = x.shape
    V = D * H * W
    dtype = dtype or x.dtype
    out = torch.empty(B, C, D, H, W, dtype=dtype, device=x.device)
    L.check(L.load().passion_copy_strided(_p(x), L.dtype_code(x.dtype), V * C, C, 1, _p(out), L.dtype_code(dtype),
                                          C * V, 1, V, B, V, C, L.stream_ptr()), "passion_copy_strided")
    return out


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
class WceDiceFn(torch.autograd.Function):
    """(weighted CE, Dice) per sample -- softmax_weighted_loss_bs + dice_loss_bs
    (reference criterions.py:59-76, 25-38) in one pass over the probabilities."""

    @staticmethod
    def forward(ctx, probs, target_cl, layout):
        L.require_gpu(probs, target_cl)
        probs = probs.contiguous()
        assert probs.dtype == torch.float32 and target_cl.dtype == torch.float32
        B, V, C, ps = _strides(probs, layout)
        lib = L.load()
        scratch = torch.empty(lib.passion_wce_dice_sums_floats(B), dtype=torch.float32, device=probs.device)
        sums = scratch[:B * 16].view(B, 16)            # the rest: the reduction's per-block slots (fixed-order sums)
        out = torch.empty(B, 2, dtype=torch.float32, device=probs.device)
        L.check(lib.passion_wce_dice_fwd(_p(probs), *ps, _p(target_cl), B, target_cl.shape[0], V, C, _p(scratch), _p(out),
                                         L.stream_ptr()), "passion_wce_dice_fwd")
        ctx.layout = layout
        ctx.save_for_backward(probs, target_cl, sums)
        return out

    @staticmethod
    def backward(ctx, g):
        probs, target_cl, sums = ctx.saved_tensors
        B, V, C, ps = _strides(probs, ctx.layout)
        g = g.contiguous().float()
        dp = torch.empty_like(probs)
        L.check(L.load().passion_wce_dice_bwd(_p(probs), *ps, _p(target_cl), _p(sums), _p(g), _p(dp), B, target_cl.shape[0], V, C,
                                              L.stream_ptr()), "passion_wce_dice_bwd")
        return dp, None, None


def wce_dice(probs, target_cl, layout="cl"):
    """-> (B,2): [:,0] weighted cross-entropy, [:,1] Dice loss"""
    return WceDiceFn.apply(probs, target_cl, layout)


class SoftmaxWceDiceFn(torch.autograd.Function):
    """wce_dice(softmax_classes(logits, inv_temp), target) without the probabilities in memory: decoder_sep's predictions (reference
    mmformer.py:112-113 nn.Softmax, :546-549 the two losses) have no other reader in training.  Bit for bit the two-op form."""

    @staticmethod
    def forward(ctx, logits, target_cl, inv_temp):
        L.require_gpu(logits, target_cl)
        logits = logits.contiguous()
        assert logits.dtype == torch.float32 and target_cl.dtype == torch.float32
        B, C = logits.shape[0], logits.shape[-1]
        V = logits.numel() // (B * C)
        lib = L.load()
        scratch = torch.empty(lib.passion_wce_dice_sums_floats(B), dtype=torch.float32, device=logits.device)
        sums = scratch[:B * 16].view(B, 16)
        out = torch.empty(B, 2, dtype=torch.float32, device=logits.device)
        L.check(lib.passion_softmax_wce_dice_fwd(_p(logits), _p(target_cl), B, target_cl.shape[0], V, C, float(inv_temp), _p(scratch), _p(out),
                                                 L.stream_ptr()), "passion_softmax_wce_dice_fwd")
        ctx.inv_temp = float(inv_temp)
        ctx.save_for_backward(logits, target_cl, sums)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, target_cl, sums = ctx.saved_tensors
        B, C = logits.shape[0], logits.shape[-1]
        V = logits.numel() // (B * C)
        g = g.contiguous().float()
        dl = torch.empty_like(logits)
        L.check(L.load().passion_softmax_wce_dice_bwd(_p(logits), _p(target_cl), _p(sums), _p(g), _p(dl), B, target_cl.shape[0], V, C,
                                                      ctx.inv_temp, L.stream_ptr()), "passion_softmax_wce_dice_bwd")
        return dl, None, None


def softmax_wce_dice(logits, target_cl, inv_temp=1.0):
    """logits f32 channels-last (B, D, H, W, 4) -> (B, 2) = (weighted CE, Dice) of softmax(logits * inv_temp) against target_cl"""
    return SoftmaxWceDiceFn.apply(logits, target_cl, inv_temp)


# decoder_sep's loss from its logits (one pass forward, one backward) instead of softmax -> probabilities -> losses; =0: the two-op form
SEP_LOSS_FUSED = __import__("os").environ.get("PASSION_SEP_LOSS_FUSED", "1") != "0"


def sep_wce_dice(sep_out, target_cl):
    """(B', 2) losses of decoder_sep's output against the label volumes; sep_out = its logits when SEP_LOSS_FUSED (the decoders are
    called with probs=not SEP_LOSS_FUSED), else its probabilities"""
    return softmax_wce_dice(sep_out, target_cl, 1.0) if SEP_LOSS_FUSED else wce_dice(sep_out, target_cl, "cl")


class KLFn(torch.autograd.Function):
    """temp_kl_loss_bs (reference criterions.py:92-103) on probabilities that were already
    softmaxed with 1/T (and upsampled).  Teacher side carries no gradient (it is detached upstream)."""

    @staticmethod
    def forward(ctx, ps, pt, temp):
        L.require_gpu(ps, pt)
        ps, pt = ps.contiguous(), pt.contiguous()
        B, D, H, W, C = ps.shape
        Bt = pt.shape[0]                 # fewer teacher samples than students: student b is held against teacher b % Bt
        assert B % Bt == 0 and pt.shape[1:] == ps.shape[1:]
        out = torch.empty(B, dtype=torch.float32, device=ps.device)
        lib = L.load()
        work = torch.empty(lib.passion_kl_work_floats(B), dtype=torch.float32, device=ps.device)
        L.check(lib.passion_kl_fwd(_p(ps), _p(pt), B, Bt, D * H * W, C, float(temp), _p(out), _p(work), L.stream_ptr()),
                "passion_kl_fwd")
        ctx.temp = float(temp)
        ctx.save_for_backward(ps, pt)
        return out.unsqueeze(1)

    @staticmethod
    def backward(ctx, g):
        ps, pt = ctx.saved_tensors
        B, D, H, W, C = ps.shape
        g = g.contiguous().float().view(-1)
        dps = torch.empty_like(ps)
        L.check(L.load().passion_kl_bwd(_p(ps), _p(pt), _p(g), _p(dps), B, pt.shape[0], D * H * W, C, ctx.temp, L.stream_ptr()),
                "passion_kl_bwd")
        return dps, None, None


def kl_probs(ps, pt, temp):
    return KLFn.apply(ps, pt.detach(), temp)


UP_FUSED = os.environ.get("PASSION_LOSS_UP_FUSED", "1") != "0"      # x k upsampling inside the loss kernels (A/B: 0 = materialise)


def _up_ok(coarse, k):
    """the fused kernels take f32 4-class maps on the GPU whose x k volume has < 2^23 voxels"""
    B, D, H, W, C = coarse.shape
    return UP_FUSED and k > 1 and coarse.is_cuda and C == 4 and coarse.dtype == torch.float32 and D * H * W * k ** 3 < (1 << 23)


class WceDiceUpFn(torch.autograd.Function):
    """(weighted CE, Dice) of nn.Upsample(x k)(coarse probabilities) against the full-size label -- the deep-supervision terms of
    reference mmformer.py:540-544 -- without the x k map: the kernels blend the coarse neighbours in registers, the backward runs
    the separable adjoint with its first (W) pass fused into the loss gradient."""

    @staticmethod
    def forward(ctx, probs_c, target_cl, k):
        L.require_gpu(probs_c, target_cl)
        probs_c = probs_c.contiguous()
        B, D, H, W, C = probs_c.shape
        lib = L.load()
        scratch = torch.empty(lib.passion_wce_dice_sums_floats(B), dtype=torch.float32, device=probs_c.device)
        sums = scratch[:B * 16].view(B, 16)
        out = torch.empty(B, 2, dtype=torch.float32, device=probs_c.device)
        L.check(lib.passion_wce_dice_up_fwd(_p(probs_c), _p(target_cl), B, target_cl.shape[0], D, H, W, int(k), _p(scratch), _p(out),
                                            L.stream_ptr()), "passion_wce_dice_up_fwd")
        ctx.k = int(k)
        ctx.save_for_backward(probs_c, target_cl, sums)
        return out

    @staticmethod
    def backward(ctx, g):
        probs_c, target_cl, sums = ctx.saved_tensors
        B, D, H, W, C = probs_c.shape
        g = g.contiguous().float()
        lib = L.load()
        dp = torch.empty_like(probs_c)
        scratch = torch.empty(lib.passion_up_loss_bwd_scratch_floats(B, D, H, W, ctx.k), dtype=torch.float32, device=g.device)
        L.check(lib.passion_wce_dice_up_bwd(_p(probs_c), _p(target_cl), _p(sums), _p(g), _p(dp), B, target_cl.shape[0], D, H, W, ctx.k,
                                            _p(scratch), L.stream_ptr()), "passion_wce_dice_up_bwd")
        return dp, None, None


def wce_dice_up(probs_coarse, target_cl, k):
    """-> (B,2) of the x k upsampled coarse probabilities (channels-last f32); falls back to upsample + wce_dice where the fused
    kernels do not apply"""
    if not _up_ok(probs_coarse, k):
        return wce_dice(upsample(probs_coarse, k), target_cl, "cl")
    return WceDiceUpFn.apply(probs_coarse, target_cl, k)


class KLUpFn(torch.autograd.Function):
    """temp_kl_loss_bs (reference criterions.py:92-103) of x k upsampled student probabilities against the x k upsampled teacher's,
    both given as COARSE maps (softmaxed with 1 / T)"""

    @staticmethod
    def forward(ctx, ps_c, pt_c, temp, k):
        L.require_gpu(ps_c, pt_c)
        ps_c, pt_c = ps_c.contiguous(), pt_c.contiguous()
        B, D, H, W, C = ps_c.shape
        Bt = pt_c.shape[0]
        assert B % Bt == 0 and pt_c.shape[1:] == ps_c.shape[1:]
        lib = L.load()
        out = torch.empty(B, dtype=torch.float32, device=ps_c.device)
        work = torch.empty(lib.passion_kl_work_floats(B), dtype=torch.float32, device=ps_c.device)
        L.check(lib.passion_kl_up_fwd(_p(ps_c), _p(pt_c), B, Bt, D, H, W, int(k), float(temp), _p(out), _p(work), L.stream_ptr()),
                "passion_kl_up_fwd")
        ctx.temp, ctx.k = float(temp), int(k)
        ctx.save_for_backward(ps_c, pt_c)
        return out.unsqueeze(1)

    @staticmethod
    def backward(ctx, g):
        ps_c, pt_c = ctx.saved_tensors
        B, D, H, W, C = ps_c.shape
        g = g.contiguous().float().view(-1)
        lib = L.load()
        dps = torch.empty_like(ps_c)
        scratch = torch.empty(lib.passion_up_loss_bwd_scratch_floats(B, D, H, W, ctx.k), dtype=torch.float32, device=g.device)
        L.check(lib.passion_kl_up_bwd(_p(ps_c), _p(pt_c), _p(g), _p(dps), B, pt_c.shape[0], D, H, W, ctx.k, ctx.temp, _p(scratch),
                                      L.stream_ptr()), "passion_kl_up_bwd")
        return dps, None, None, None


def kl_probs_up(ps_coarse, pt_coarse, temp, k):
    """KL of the x k upsampled maps; pt_coarse carries no gradient"""
    if not _up_ok(ps_coarse, k):
        return kl_probs(upsample(ps_coarse, k), upsample(pt_coarse.detach(), k), temp)
    return KLUpFn.apply(ps_coarse, pt_coarse.detach(), temp, k)


class ProtoFn(torch.autograd.Function):
    """prototype_passion_loss_bs (reference criterions.py:144-180): per-class masked-mean prototypes,
    cosine-similarity maps of student and teacher, mean squared / absolute difference."""

    @staticmethod
    def forward(ctx, fs, ft, target_cl):
        L.require_gpu(fs, ft, target_cl)
        fs, ft = fs.contiguous(), ft.contiguous()
        B, D, H, W, Cf = fs.shape
        lib = L.load()
        work = torch.empty(lib.passion_proto_work_floats(B, Cf), dtype=torch.float32, device=fs.device)
        out = torch.empty(B, 2, dtype=torch.float32, device=fs.device)
        Bt = ft.shape[0]                 # the four students batched against one teacher: sample b uses ft / target of b % Bt
        assert B % Bt == 0 and target_cl.shape[0] == Bt
        L.check(lib.passion_proto_fwd(_p(fs), _p(ft), L.dtype_code(fs.dtype), _p(target_cl), B, Bt, D * H * W, Cf,
                                      _p(work), _p(out), L.stream_ptr()), "passion_proto_fwd")
        ctx.save_for_backward(fs, ft, target_cl, work)
        loss, dist = out[:, 0:1].clone(), out[:, 1:2].clone()
        ctx.mark_non_differentiable(dist)
        return loss, dist

    @staticmethod
    def backward(ctx, g, _gd):
        fs, ft, target_cl, work = ctx.saved_tensors
        B, D, H, W, Cf = fs.shape
        g = g.contiguous().float().view(-1)
        dfs = torch.empty_like(fs)
        L.check(L.load().passion_proto_bwd(_p(fs), _p(ft), L.dtype_code(fs.dtype), _p(target_cl), _p(g), B, ft.shape[0],
                                           D * H * W, Cf, _p(work), _p(dfs), L.stream_ptr()), "passion_proto_bwd")
        return dfs, None, None


def proto_loss(fs, ft, target_cl):
    return ProtoFn.apply(fs, ft.detach(), target_cl)


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _draw_seed():
    """64-bit seed of an attention-dropout mask from torch's CPU generator (follows torch.manual_seed; no device sync)"""
    return int(torch.empty((), dtype=torch.int64).random_().item()) & (2 ** 64 - 1)


class TokenDropFn(torch.autograd.Function):
    """out = [x +] dropout(dropout(act(y))) in one launch each way (csrc/token_ops.hip token_drop_*): the glue of the reference's
    Transformer blocks -- `x + self.dropout(self.proj_drop(proj))` (mmformer.py:226,247,256), `dropout(gelu(h))` (:271-273),
    `x + dropout(h)` (:275,247).  y: GEMM output (bf16 under autocast, else f32); x: the f32 residual stream or None;
    ndrop: nn.Dropout modules in a row (independent masks); act: 0 identity, 1 GELU.  The masks are regenerated in the backward from the
    forward's seed (drawn from torch's generator: follows torch.manual_seed)."""

    @staticmethod
    def forward(ctx, y, x, p, ndrop, act):
        L.require_gpu(y)
        y = y.contiguous()
        out_dtype = torch.float32 if x is not None else y.dtype
        out = torch.empty(y.shape, dtype=out_dtype, device=y.device)
        seed = _draw_seed() if p > 0 else 0
        xs = x.contiguous() if x is not None else None
        L.check(L.load().passion_token_drop_fwd(_p(y), _p(xs), _p(out), y.numel(), L.dtype_code(y.dtype), L.dtype_code(out_dtype), float(p), ndrop, act,
                                                seed, L.stream_ptr()), "passion_token_drop_fwd")
        ctx.meta = (float(p), ndrop, act, seed, x is not None)
        ctx.save_for_backward(y if act else y.new_empty(0))
        ctx.ydtype, ctx.yshape = y.dtype, y.shape
        return out

    @staticmethod
    def backward(ctx, g):
        p, ndrop, act, seed, has_x = ctx.meta
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        dy = torch.empty(ctx.yshape, dtype=ctx.ydtype, device=g.device)
        L.check(L.load().passion_token_drop_bwd(_p(g), _p(y) if act else c_void_p(0), _p(dy), dy.numel(), L.dtype_code(ctx.ydtype), L.dtype_code(g.dtype),
                                                p, ndrop, act, seed, L.stream_ptr()), "passion_token_drop_bwd")
        return dy, (g if has_x else None), None, None, None


def token_drop_ok(y, x=None):
    """the fused kernel takes GPU tensors of 8-element multiples in (bf16 | f32) with an f32 residual"""
    return (y.is_cuda and y.dtype in (torch.bfloat16, torch.float32) and y.numel() % 8 == 0 and y.numel() < 2 ** 32 and
            (x is None or (x.dtype == torch.float32 and x.shape == y.shape)))


def token_drop(y, x=None, p=0.0, ndrop=1, act=0):
    """[x +] dropout^ndrop(act(y)) (training-mode dropout with probability p each)"""
    return TokenDropFn.apply(y, x, float(p), int(ndrop), int(act))


class FlashAttentionFn(torch.autograd.Function):
    """fused attention for bf16, 64-wide heads (csrc/flash_attn.hip): no (B,H,N,N) tensor, backward recomputes from the row
    log-sum-exp; attention dropout is a counter-based mask regenerated in the backward from the forward's seed"""

    @staticmethod
    def forward(ctx, qkv, heads, scale, drop_p):
        L.require_gpu(qkv)
        qkv = qkv.contiguous()
        B, N, C3 = qkv.shape
        Dh = C3 // 3 // heads
        seed = _draw_seed() if drop_p > 0.0 else 0
        out = torch.empty(B, N, heads * Dh, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, heads, N, dtype=torch.float32, device=qkv.device)
        L.check(L.load().passion_flash_attn_fwd(_p(qkv), B, N, heads, Dh, float(scale), float(drop_p), seed, _p(out), _p(lse),
                                                L.stream_ptr()), "passion_flash_attn_fwd")
        ctx.heads, ctx.scale, ctx.drop_p, ctx.seed = heads, float(scale), float(drop_p), seed
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        B, N, C3 = qkv.shape
        H = ctx.heads
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
        L.check(L.load().passion_flash_attn_bwd(_p(qkv), _p(out), _p(dout), _p(lse), B, N, H, C3 // 3 // H, ctx.scale, ctx.drop_p,
                                                ctx.seed, _p(delta), _p(dqkv), L.stream_ptr()), "passion_flash_attn_bwd")
        return dqkv, None, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T * scale) v for qkv (B,N,3,H,Dh) -> (B,N,H*Dh)  (reference mmformer.py:220-224), general path: f32 parity
    mode (exact f32 MFMA) and head widths other than 64; the scores are materialised.
    `drop_p` > 0 applies attention dropout to the probabilities (mmformer.py:222)."""

    @staticmethod
    def forward(ctx, qkv, heads, scale, drop_p):
        L.require_gpu(qkv)
        qkv = qkv.contiguous()
        B, N, C3 = qkv.shape
        Dh = C3 // 3 // heads
        lib = L.load()
        dt = L.dtype_code(qkv.dtype)
        probs = torch.empty(B, heads, N, N, dtype=torch.float32, device=qkv.device)
        L.check(lib.passion_attn_scores_fwd(_p(qkv), dt, B, N, heads, Dh, float(scale), _p(probs), L.stream_ptr()),
                "passion_attn_scores_fwd")
        pd, seed = probs, 0
        if drop_p > 0.0:     # one pass, mask regenerated from the seed in backward (no rand / compare / mask tensors)
            seed = _draw_seed()
            pd = torch.empty_like(probs)
            L.check(lib.passion_dropout(_p(probs), _p(pd), probs.numel(), float(drop_p), seed, L.stream_ptr()),
                    "passion_dropout")
        out = torch.empty(B, N, heads * Dh, dtype=qkv.dtype, device=qkv.device)
        L.check(lib.passion_attn_pv_fwd(_p(pd), _p(qkv), dt, B, N, heads, Dh, _p(out), L.stream_ptr()),
                "passion_attn_pv_fwd")
        ctx.heads, ctx.scale, ctx.drop_p, ctx.seed = heads, float(scale), drop_p, seed
        ctx.save_for_backward(qkv, probs, pd if drop_p > 0.0 else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, probs, pd = ctx.saved_tensors
        B, N, C3 = qkv.shape
        H = ctx.heads
        Dh = C3 // 3 // H
        lib = L.load()
        dt = L.dtype_code(qkv.dtype)
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dprobs = torch.empty_like(probs)
        L.check(lib.passion_attn_pv_bwd(_p(pd if pd is not None else probs), _p(qkv), _p(dout), dt, B, N, H, Dh,
                                        _p(dprobs), _p(dqkv), L.stream_ptr()), "passion_attn_pv_bwd")
        if pd is not None:           # the same mask, in place
            L.check(lib.passion_dropout(_p(dprobs), _p(dprobs), dprobs.numel(), float(ctx.drop_p), ctx.seed,
                                        L.stream_ptr()), "passion_dropout")
        L.check(lib.passion_attn_scores_bwd(_p(probs), _p(dprobs), _p(qkv), dt, B, N, H, Dh, ctx.scale, _p(dqkv),
                                            L.stream_ptr()), "passion_attn_scores_bwd")
        return dqkv, None, None, None


class MaskedAttentionFn(torch.autograd.Function):
    """MaskedAttention core of the m2ftrans bottleneck (reference m2ftrans.py:336-349): softmax over the keys the block mask
    allows (mask.py:5-22), attention dropout, @ v.  qkv (B,N,3,H,Dh), allow (B,G,G) uint8 over G groups of N / G tokens.
    -> (out (B,N,H*Dh), attn (B,H,N,N) f32 = the probabilities AFTER dropout, which the reference hands on detached (:380))."""

    @staticmethod
    def forward(ctx, qkv, allow, heads, scale, drop_p):
        L.require_gpu(qkv, allow)
        qkv = qkv.contiguous()
        allow = allow.contiguous()
        B, N, C3 = qkv.shape
        Dh = C3 // 3 // heads
        G = allow.shape[1]
        assert allow.dtype == torch.uint8 and allow.shape == (B, G, G) and N % G == 0
        lib = L.load()
        dt = L.dtype_code(qkv.dtype)
        probs = torch.empty(B, heads, N, N, dtype=torch.float32, device=qkv.device)
        L.check(lib.passion_attn_scores_masked_fwd(_p(qkv), dt, B, N, heads, Dh, float(scale), _p(allow), G, _p(probs), L.stream_ptr()),
                "passion_attn_scores_masked_fwd")
        pd, seed = probs, 0
        if drop_p > 0.0:
            seed = _draw_seed()
            pd = torch.empty_like(probs)
            L.check(lib.passion_dropout(_p(probs), _p(pd), probs.numel(), float(drop_p), seed, L.stream_ptr()), "passion_dropout")
        out = torch.empty(B, N, heads * Dh, dtype=qkv.dtype, device=qkv.device)
        L.check(lib.passion_attn_pv_fwd(_p(pd), _p(qkv), dt, B, N, heads, Dh, _p(out), L.stream_ptr()), "passion_attn_pv_fwd")
        ctx.heads, ctx.scale, ctx.drop_p, ctx.seed = heads, float(scale), drop_p, seed
        ctx.save_for_backward(qkv, probs, pd if drop_p > 0.0 else None)
        ctx.mark_non_differentiable(pd)
        return out, pd

    @staticmethod
    def backward(ctx, dout, _dattn):
        qkv, probs, pd = ctx.saved_tensors
        B, N, C3 = qkv.shape
        H = ctx.heads
        Dh = C3 // 3 // H
        lib = L.load()
        dt = L.dtype_code(qkv.dtype)
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dprobs = torch.empty_like(probs)
        L.check(lib.passion_attn_pv_bwd(_p(pd if pd is not None else probs), _p(qkv), _p(dout), dt, B, N, H, Dh, _p(dprobs), _p(dqkv),
                                        L.stream_ptr()), "passion_attn_pv_bwd")
        if pd is not None:
            L.check(lib.passion_dropout(_p(dprobs), _p(dprobs), dprobs.numel(), float(ctx.drop_p), ctx.seed, L.stream_ptr()),
                    "passion_dropout")
        L.check(lib.passion_attn_scores_bwd(_p(probs), _p(dprobs), _p(qkv), dt, B, N, H, Dh, ctx.scale, _p(dqkv), L.stream_ptr()),
                "passion_attn_scores_bwd")
        return dqkv, None, None, None, None


def masked_attention(qkv, allow, heads, scale, drop_p=0.0):
    return MaskedAttentionFn.apply(qkv, allow, heads, scale, drop_p)


_F32_STACKS = {}


def f32_stack(params):
    """G same-shaped f32 parameters back to back in one (G, *shape) buffer, re-stacked only when one of them changed"""
    if len(params) == 1:
        return params[0].detach().unsqueeze(0)
    key = tuple(id(p) for p in params)
    ver = tuple((p.data_ptr(), p._version) for p in params)
    ent = _F32_STACKS.get(key)
    if ent is None or ent[1] != ver or ent[0].device != params[0].device:
        want = (len(params), *params[0].shape)             # ids are reused once a model is gone: the old buffer may have another shape
        buf = ent[0] if ent is not None and ent[0].device == params[0].device and tuple(ent[0].shape) == want else None
        if buf is None:
            buf = torch.stack([p.detach() for p in params])
        else:                                              # ONE multi-tensor launch (torch.stack(out=) issues a device copy per parameter:
            torch._foreach_copy_(list(buf.unbind(0)), [p.detach() for p in params])        # 103 per m2ftrans step, weights change every step)
        ent = _F32_STACKS[key] = (buf, ver)
    return ent[0]


class LnActFn(torch.autograd.Function):
    """y = act(LayerNorm_C(x) * gamma + beta [+ res]) for G row groups with their own (gamma, beta): the LayerNorm -> GELU pairs
    and the LayerNorm -> residual -> GELU tail of DepthWiseConvBlock / GroupConvBlock (reference blocks.py:32-109), one launch
    forward, two backward (csrc/token_ops.hip).  x: (G, R, C) or any (..., C) with G = 1; params: G gammas then G betas."""

    @staticmethod
    def forward(ctx, x, res, act, eps, G, out_dtype, *params):
        L.require_gpu(x)
        ctx.box = None
        if len(params) == 1 and isinstance(params[0], _Box):      # direct-gradient mode
            ctx.box = params[0]
            params = ctx.box.params
        x = x.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        gam, bet = f32_stack(params[:G]), f32_stack(params[G:])
        if res is not None:
            res = res.contiguous()
            assert res.shape == x.shape and res.dtype == x.dtype
        out_dtype = x.dtype if out_dtype is None else out_dtype     # bf16 out of an f32 input: the cast the next GEMM would launch
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        stat = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
        L.check(L.load().passion_ln_act_fwd(_p(x), _p(gam), _p(bet), _p(res), _p(y), _p(stat), L.dtype_code(x.dtype), L.dtype_code(out_dtype),
                                            rows, C, rows // G, float(eps), int(act), L.stream_ptr()), "passion_ln_act_fwd")
        ctx.save_for_backward(x, res if act else None, stat, gam, bet)
        ctx.act, ctx.G, ctx.has_res, ctx.out_dtype = int(act), G, res is not None, out_dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, stat, gam, bet = ctx.saved_tensors
        G, C = ctx.G, x.shape[-1]
        rows = x.numel() // C
        dy = dy.contiguous()
        if dy.dtype != ctx.out_dtype:
            dy = dy.to(ctx.out_dtype)
        lib = L.load()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.act) else None
        part = torch.empty(lib.passion_ln_act_part_floats(rows, C, rows // G), dtype=torch.float32, device=x.device)
        dgb = torch.empty(2, G, C, dtype=torch.float32, device=x.device)
        L.check(lib.passion_ln_act_bwd(_p(x), _p(gam), _p(bet), _p(res), _p(stat), _p(dy), _p(dx), _p(dres), _p(part), _p(dgb[0]), _p(dgb[1]),
                                       L.dtype_code(x.dtype), L.dtype_code(ctx.out_dtype), rows, C, rows // G, ctx.act, L.stream_ptr()),
                "passion_ln_act_bwd")
        if ctx.has_res and not ctx.act:
            dres = dy
        if ctx.box is not None:
            for g in range(G):
                _give(ctx.box.params[g], dgb[0, g])
                _give(ctx.box.params[G + g], dgb[1, g])
            return dx, dres, None, None, None, None, None
        return (dx, dres, None, None, None, None, *[dgb[0, g] for g in range(G)], *[dgb[1, g] for g in range(G)])


def ln_act(x, norms, act=False, res=None, out_dtype=None):
    """norms: one nn.LayerNorm or a list of G (x then is (G, R, C), group g normalised with norms[g]'s affine).
    out_dtype=torch.bfloat16 with an f32 x (no res): the result leaves in bf16, what the GEMM behind it would cast it to"""
    if not isinstance(norms, (list, tuple)):
        norms = [norms]
    G, C = len(norms), x.shape[-1]
    rows = x.numel() // C
    assert G == 1 or x.shape[0] == G
    if not (x.is_cuda and L.load().passion_ln_act_supported(rows, C, rows // G)):
        raise RuntimeError(f"ln_act: unsupported shape rows={rows} C={C} G={G} (C / 8 must be a power of two, C <= 512) or not on the GPU")
    params = [n.weight for n in norms] + [n.bias for n in norms]
    if out_dtype is not None and (out_dtype == x.dtype or res is not None or x.dtype != torch.float32):
        out_dtype = None
    if _direct(x, params):
        return LnActFn.apply(x, res, bool(act), norms[0].eps, G, out_dtype, _Box(params))
    return LnActFn.apply(x, res, bool(act), norms[0].eps, G, out_dtype, *params)


class GroupConvFn(torch.autograd.Function):
    """grouped 3x3x3 Conv3d (zero padding, stride 1, in = out channels, `cpg` channels per group) on channels-last tensors:
    DepthWiseConvBlock.conv2 / GroupConvBlock.dwconv of the m2ftrans cross-token blocks (reference blocks.py:45-47, 77-79).
    G weight sets in one launch: x is (G * B, D, H, W, C) and samples [g B, (g + 1) B) use set g; params: G weights then G biases."""

    @staticmethod
    def forward(ctx, x, G, has_bias, *params):
        L.require_gpu(x)
        ctx.box = None
        if len(params) == 1 and isinstance(params[0], _Box):      # direct-gradient mode
            ctx.box = params[0]
            params = ctx.box.params
        x = x.contiguous()
        B, D, H, W, C = x.shape
        w = f32_stack(params[:G]).contiguous()
        cpg = w.shape[2]
        assert w.shape == (G, C, cpg, 3, 3, 3) and w.dtype == torch.float32 and B % G == 0
        b32 = f32_stack(params[G:]).contiguous() if has_bias else None
        y = torch.empty_like(x)
        L.check(L.load().passion_gconv3d_fwd(_p(x), _p(w), _p(b32), _p(y), L.dtype_code(x.dtype), B, D, H, W, C, cpg, G, L.stream_ptr()),
                "passion_gconv3d_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.G = has_bias, G
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, D, H, W, C = x.shape
        G, cpg = ctx.G, w.shape[2]
        dy = dy.contiguous()
        lib = L.load()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(G, C, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        n = lib.passion_gconv3d_part_floats(B, D, H, W, C, cpg, G)
        part = torch.empty(n, dtype=torch.float32, device=x.device) if n else None
        L.check(lib.passion_gconv3d_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), _p(part), L.dtype_code(x.dtype), B, D, H, W, C, cpg, G,
                                        L.stream_ptr()), "passion_gconv3d_bwd")
        if ctx.box is not None:
            for g in range(G):
                _give(ctx.box.params[g], dw[g])
                if ctx.has_bias:
                    _give(ctx.box.params[G + g], db[g])
            return dx, None, None, None
        return (dx, None, None, *[dw[g] for g in range(G)], *([db[g] for g in range(G)] if ctx.has_bias else []))


def group_conv3d(x, weight, bias):
    """weight / bias: one parameter each, or lists of G (x then holds G * B samples, B per weight set)"""
    ws = list(weight) if isinstance(weight, (list, tuple)) else [weight]
    bs = (list(bias) if isinstance(bias, (list, tuple)) else [bias]) if bias is not None else []
    if _direct(x, ws + bs):
        return GroupConvFn.apply(x, len(ws), bool(bs), _Box(ws + bs))
    return GroupConvFn.apply(x, len(ws), bool(bs), *ws, *bs)


class ScaleNearestFn(torch.autograd.Function):
    """y[b] = x[b % Bx] * nearest_upsample(w[b], 2^k): Weight_Attention of m2ftrans (reference m2ftrans.py:443-458); x (Bx,D,H,W,C),
    w (Bv, D >> k, H >> k, W >> k) f32 without gradient (the attention maps are detached), Bv / Bx masked passes in one launch"""

    @staticmethod
    def forward(ctx, x, w, k):
        L.require_gpu(x, w)
        x, w = x.contiguous(), w.contiguous().float()
        Bx, D, H, W, C = x.shape
        Bv = w.shape[0]
        assert Bv % Bx == 0 and w.shape[1:] == (D >> k, H >> k, W >> k)
        y = torch.empty(Bv, D, H, W, C, dtype=x.dtype, device=x.device)
        L.check(L.load().passion_scale_nearest_fwd(_p(x), _p(w), _p(y), L.dtype_code(x.dtype), Bv, Bx, D, H, W, C, k, L.stream_ptr()),
                "passion_scale_nearest_fwd")
        ctx.k, ctx.shape = k, x.shape
        ctx.save_for_backward(w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        Bx, D, H, W, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        L.check(L.load().passion_scale_nearest_bwd(_p(dy), _p(w), _p(dx), L.dtype_code(dy.dtype), w.shape[0], Bx, D, H, W, C, ctx.k,
                                                   L.stream_ptr()), "passion_scale_nearest_bwd")
        return dx, None, None


def scale_nearest(x, w, k):
    return ScaleNearestFn.apply(x, w.detach(), k)


FLASH = True      # A/B switch (tests compare the two paths)


def attention(qkv, heads, scale, drop_p=0.0):
    if FLASH and qkv.dtype == torch.bfloat16 and qkv.shape[-1] // 3 // heads == 64:
        return FlashAttentionFn.apply(qkv, heads, scale, drop_p)
    return AttentionFn.apply(qkv, heads, scale, drop_p)


# ------------------------------------------------------------------------------------------------
# RFNet ops: post-norm materialisation and region-aware modal fusion
# ------------------------------------------------------------------------------------------------
class NormActFn(torch.autograd.Function):
    """sum_k LeakyReLU(InstanceNorm(x_k)) for 1 or 2 raw conv outputs -- the value a post-norm
    general_conv3d (reference blocks.py:366-370) hands to a non-conv consumer."""

    @staticmethod
    def forward(ctx, slope, x0, st0, x1, st1):
        L.require_gpu(x0)
        x0 = x0.contiguous()
        B, D, H, W, C = x0.shape
        out = torch.empty_like(x0)
        L.check(L.load().passion_norm_act_fwd(_p(x0), _p(st0), _p(x1), _p(st1), _p(out), L.dtype_code(x0.dtype), B,
                                              D * H * W, C, C, 0, float(slope), L.stream_ptr()), "passion_norm_act_fwd")
        ctx.slope = float(slope)
        ctx.save_for_backward(x0, st0, x1, st1)
        return out

    @staticmethod
    def backward(ctx, dout):
        x0, st0, x1, st1 = ctx.saved_tensors
        B, D, H, W, C = x0.shape
        V = D * H * W
        lib = L.load()
        dout = dout.contiguous()
        scratch = torch.empty(lib.passion_norm_act_scratch_floats(B, V, C), dtype=torch.float32, device=x0.device)
        outs = []
        for x, st in ((x0, st0), (x1, st1)):
            if x is None:
                outs.append(None)
                continue
            dx = torch.empty_like(x)
            L.check(lib.passion_norm_act_bwd(_p(x), _p(st), _p(dout), _p(dx), L.dtype_code(x.dtype), B, V, C, C, 0,
                                             ctx.slope, _p(scratch), L.stream_ptr()), "passion_norm_act_bwd")
            outs.append(dx)
        return None, outs[0], None, outs[1], None


def norm_act(x0, st0, x1=None, st1=None, slope=0.2):
    return NormActFn.apply(slope, x0, st0, x1, st1)


def _ptr4(ts):
    arr = (c_void_p * 4)()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


class RegionGapFn(torch.autograd.Function):
    """S[b,i,m,c] = sum_v y_m[c] prob_i and P[b,i] = sum_v prob_i -- the global average pools of
    modal_fusion (reference blocks.py:507-508) for all regions x modalities in one pass.
    prob / mask may hold P*B rows (P masked passes over the same B samples of y, see SegIn.bmod)."""

    @staticmethod
    def forward(ctx, prob, mask, y0, y1, y2, y3):
        ys = [t.contiguous() for t in (y0, y1, y2, y3)]
        L.require_gpu(prob, *ys)
        Bp, D, H, W, C = ys[0].shape
        Bv = prob.shape[0]
        bmod = Bp if Bv != Bp else 0
        V = D * H * W
        lib = L.load()
        scratch = torch.empty(lib.passion_region_scratch_floats(Bv, V, C), dtype=torch.float32, device=prob.device)
        sums = torch.empty(Bv, 16 * C + 4, dtype=torch.float32, device=prob.device)
        L.check(lib.passion_region_gap_fwd(_ptr4(ys), _p(mask), mask.shape[1] if mask is not None else 0, _p(prob),
                                           L.dtype_code(ys[0].dtype), Bv, V, C, bmod, _p(scratch), _p(sums), L.stream_ptr()),
                "passion_region_gap_fwd")
        ctx.save_for_backward(prob, mask)
        ctx.meta = (Bv, V, C, ys[0].dtype, ys[0].shape, bmod)
        S = sums[:, :16 * C].reshape(Bv, 4, 4, C)
        P = sums[:, 16 * C:]
        ctx.mark_non_differentiable(P)
        return S, P

    @staticmethod
    def backward(ctx, gS, _gP):
        prob, mask = ctx.saved_tensors
        Bv, V, C, dt, shape, bmod = ctx.meta
        gS = gS.contiguous().float()
        # bmod: the passes' gradients add up -- inside the kernel (f32 sum over the passes, written once in the shape of y)
        dys = [torch.empty(tuple(shape), dtype=dt, device=prob.device) if ctx.needs_input_grad[2 + m] else None for m in range(4)]
        L.check(L.load().passion_region_gap_bwd(_ptr4(dys), _p(mask), mask.shape[1] if mask is not None else 0, _p(prob),
                                                _p(gS), L.dtype_code(dt), Bv, V, C, bmod, L.stream_ptr()), "passion_region_gap_bwd")
        return (None, None, *dys)


class RegionMixFn(torch.autograd.Function):
    """R[b,v,i*C+c] = prob_i(v) * sum_m w[b,i,m] y_m[c](v): region split + weighted modality sum + reshape
    (reference blocks.py:602-616, 516, 530) without the (B,4,4,C,D,H,W) intermediate."""

    @staticmethod
    def forward(ctx, prob, mask, w, y0, y1, y2, y3):
        ys = [t.contiguous() for t in (y0, y1, y2, y3)]
        Bp, D, H, W, C = ys[0].shape
        Bv = prob.shape[0]
        bmod = Bp if Bv != Bp else 0
        V = D * H * W
        w = w.contiguous().float()
        out = torch.empty(Bv, D, H, W, 4 * C, dtype=ys[0].dtype, device=prob.device)
        L.check(L.load().passion_region_mix_fwd(_ptr4(ys), _p(mask), mask.shape[1] if mask is not None else 0, _p(prob),
                                                _p(w), _p(out), L.dtype_code(ys[0].dtype), Bv, V, C, bmod, L.stream_ptr()),
                "passion_region_mix_fwd")
        ctx.save_for_backward(prob, mask, w, *ys)
        ctx.bmod = bmod
        return out

    @staticmethod
    def backward(ctx, dR):
        prob, mask, w, *ys = ctx.saved_tensors
        Bp, D, H, W, C = ys[0].shape
        Bv, bmod = prob.shape[0], ctx.bmod
        V = D * H * W
        lib = L.load()
        dR = dR.contiguous()
        vshape = (Bp, D, H, W, C)             # bmod: the sum over the passes, formed inside the kernel
        dys = [torch.empty(vshape, dtype=ys[m].dtype, device=prob.device) if ctx.needs_input_grad[3 + m] else None
               for m in range(4)]
        scratch = torch.empty(lib.passion_region_scratch_floats(Bv, V, C), dtype=torch.float32, device=prob.device)
        dw = torch.empty(Bv, 4, 4, dtype=torch.float32, device=prob.device)
        L.check(lib.passion_region_mix_bwd(_ptr4(ys), _ptr4(dys), _p(mask), mask.shape[1] if mask is not None else 0,
                                           _p(prob), _p(w), _p(dR), L.dtype_code(ys[0].dtype), Bv, V, C, bmod, _p(scratch),
                                           _p(dw), L.stream_ptr()), "passion_region_mix_bwd")
        return (None, None, dw, *dys)


def region_gap(prob, mask, ys):
    return RegionGapFn.apply(prob, mask, *ys)


def region_mix(prob, mask, w, ys):
    return RegionMixFn.apply(prob, mask, w, *ys)


# ------------------------------------------------------------------------------------------------
# kernel timing hooks (bench.py)
# ------------------------------------------------------------------------------------------------
def set_option(name, value):
    """run-time A/B switch of the library (see passion_set_option in include/passion_hip.h); returns the old value"""
    old = L.load().passion_set_option(name.encode(), int(value))
    if old < 0:
        raise ValueError(f"unknown libpassion_hip option {name!r}")
    if name == "force_f32_mfma":
        _FORCE_F32[0] = bool(int(value))
        if _FORCE_F32[0]:
            set_pack_f32(True)                   # bf16 tensors on the f32 kernels: those read the f32 images
    return old


def prof_enable(on):
    """on: False/0 = off, True = all conv kinds, or a bitmask (bit0 fwd k3, bit1 fwd k1, bit2 dgrad, bit3 wgrad)"""
    mask = 0xF if on is True else int(on or 0)
    L.load().passion_prof_enable(mask)


def prof_filter(lo=0.0, hi=1e300):
    """keep event pairs only for launches whose algorithmic bytes lie in [lo, hi] (default: all)"""
    L.load().passion_prof_filter(float(lo), float(hi))


def prof_read(cap=65536, keep=False):
    """(kind, ms, algorithmic bytes, flops, kernel name) per timed launch.  keep=True leaves the records in place: the event pairs
    recorded during a HIP-graph capture are re-timed by every replay."""
    recs = (L.ProfRec * cap)()
    lib = L.load()
    n = (lib.passion_prof_peek if keep else lib.passion_prof_read)(recs, cap)
    return [(recs[i].kind, recs[i].ms, recs[i].bytes, recs[i].flops, recs[i].name.decode()) for i in range(n)]
