"""ctypes binding of libpassion_hip.so (C ABI declared in include/passion_hip.h).

The product path has no CPU or eager fallback: if the shared library is missing, or a kernel is
asked to run on a non-GPU tensor, this module raises.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PASSION_HIP_LIB") or os.path.join(_HERE, "lib", "libpassion_hip.so")   # env: A/B builds

F32, BF16 = 0, 1
PAD_ZEROS, PAD_REFLECT = 0, 1
MAX_SEG = 4


class Seg(Structure):
    _fields_ = [("x", c_void_p), ("stats", c_void_p), ("mask", c_void_p), ("dx", c_void_p),
                ("bstride", c_int64), ("C", c_int32), ("mask_col", c_int32),
                ("mask_ld", c_int32), ("bmod", c_int32), ("dx_add", c_void_p), ("dx_add2", c_void_p)]


class ConvDesc(Structure):
    _fields_ = [("x_dtype", c_int32), ("y_dtype", c_int32), ("B", c_int32), ("D", c_int32),
                ("H", c_int32), ("W", c_int32), ("ksize", c_int32), ("stride", c_int32),
                ("pad_mode", c_int32), ("nseg", c_int32), ("Cout", c_int32),
                ("lrelu_slope", c_float), ("seg", Seg * MAX_SEG)]


ABI_VERSION = 2      # include/passion_hip.h PASSION_ABI_VERSION (2: passion_seg_t.dx_add)


class ProfRec(Structure):
    _fields_ = [("kind", c_int32), ("ms", c_float), ("bytes", c_double), ("flops", c_double), ("name", c_char * 64)]


# name -> (restype, argtypes).  Must list every symbol include/passion_hip.h declares
# (tests/test_abi.py cross-checks the header against this table and against the .so).
SIGNATURES = {
    "passion_abi_version": (c_int, []),
    "passion_instnorm_scratch_doubles": (c_int64, [c_int, c_int64, c_int]),
    "passion_instnorm_stats": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int64, c_float,
                                       c_void_p, c_void_p, c_void_p]),
    "passion_conv_pack_floats": (c_int64, [c_int, c_int, c_int]),
    "passion_conv_bwd_scratch_floats": (c_int64, [POINTER(ConvDesc)]),
    "passion_conv_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "passion_conv_pack_weights": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "passion_conv3d_fwd": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "passion_conv3d_fwd_stats": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_float, c_void_p]),
    "passion_conv3d_bwd": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "passion_conv3d_fwd_grouped": (c_int, [POINTER(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_float, c_void_p]),
    "passion_conv3d_bwd_grouped": (c_int, [POINTER(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "passion_conv3d_bwd_phase": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                         c_int, c_int, c_void_p]),
    "passion_conv3d_bwd_grouped_phase": (c_int, [POINTER(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                                 c_int, c_int, c_void_p]),
    "passion_halo_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "passion_halo_unpack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "passion_halo_extend": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p]),
    "passion_halo_extend_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "passion_stats_to_moments": (c_int, [c_void_p, c_void_p, c_int64, c_double, c_double, c_void_p]),
    "passion_moments_to_stats": (c_int, [c_void_p, c_void_p, c_int64, c_double, c_void_p]),
    "passion_upsample_window_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_void_p]),
    "passion_upsample_window_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_void_p]),
    "passion_upsample_window_bwd_scratch_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "passion_upsample_window_bwd_sep": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_int, c_void_p, c_void_p]),
    "passion_upsample_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_void_p]),
    "passion_upsample_fwd_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_float, c_void_p]),
    "passion_upsample_bwd_scratch_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "passion_upsample_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_void_p, c_void_p]),
    "passion_softmax_fwd": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                    c_int64, c_int64, c_int, c_int64, c_int, c_float, c_void_p]),
    "passion_softmax_bwd": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                    c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int,
                                    c_int64, c_int, c_float, c_void_p]),
    "passion_target_to_cl": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "passion_copy_strided": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_int,
                                     c_int64, c_int64, c_int64, c_int, c_int64, c_int, c_void_p]),
    "passion_wce_dice_fwd": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int,
                                     c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "passion_wce_dice_bwd": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p]),
    "passion_softmax_wce_dice_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "passion_softmax_wce_dice_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_float, c_void_p]),
    "passion_wce_dice_up_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "passion_wce_dice_up_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p]),
    "passion_kl_up_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "passion_kl_up_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                  c_void_p]),
    "passion_up_loss_bwd_scratch_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "passion_colsum_work_floats": (c_int64, [c_int]),
    "passion_colsum": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "passion_colsum_batched": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "passion_wce_dice_sums_floats": (c_int64, [c_int]),
    "passion_kl_work_floats": (c_int64, [c_int]),
    "passion_kl_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_float, c_void_p,
                               c_void_p, c_void_p]),
    "passion_kl_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int,
                               c_float, c_void_p]),
    "passion_proto_work_floats": (c_int64, [c_int, c_int]),
    "passion_proto_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_int,
                                  c_void_p, c_void_p, c_void_p]),
    "passion_proto_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int64,
                                  c_int, c_void_p, c_void_p, c_void_p]),
    "passion_attn_scores_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                        c_void_p, c_void_p]),
    "passion_attn_pv_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p]),
    "passion_attn_pv_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_void_p, c_void_p, c_void_p]),
    "passion_flash_attn_supported": (c_int, [c_int, c_int]),
    "passion_flash_attn_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, ctypes.c_uint64, c_void_p, c_void_p,
                                       c_void_p]),
    "passion_flash_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                                       ctypes.c_uint64, c_void_p, c_void_p, c_void_p]),
    "passion_dropout": (c_int, [c_void_p, c_void_p, c_int64, ctypes.c_float, ctypes.c_uint64, c_void_p]),
    "passion_token_drop_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, ctypes.c_float, c_int, c_int, ctypes.c_uint64, c_void_p]),
    "passion_token_drop_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, ctypes.c_float, c_int, c_int, ctypes.c_uint64, c_void_p]),
    "passion_gconv3d_part_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "passion_gconv3d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "passion_gconv3d_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p]),
    "passion_ln_act_supported": (c_int, [c_int64, c_int, c_int64]),
    "passion_ln_act_part_floats": (c_int64, [c_int64, c_int, c_int64]),
    "passion_ln_act_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int64, c_float,
                                   c_int, c_void_p]),
    "passion_ln_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_int64, c_int, c_int64, c_int, c_void_p]),
    "passion_scale_nearest_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "passion_scale_nearest_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "passion_attn_scores_masked_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "passion_attn_scores_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_int, c_float, c_void_p, c_void_p]),
    "passion_norm_act_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int,
                                     c_int, c_int, c_float, c_void_p]),
    "passion_norm_act_scratch_floats": (c_int64, [c_int, c_int64, c_int]),
    "passion_norm_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int,
                                     c_int, c_float, c_void_p, c_void_p]),
    "passion_region_scratch_floats": (c_int64, [c_int, c_int64, c_int]),
    "passion_region_gap_fwd": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_int,
                                       c_int, c_void_p, c_void_p, c_void_p]),
    "passion_region_gap_bwd": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int64,
                                       c_int, c_int, c_void_p]),
    "passion_region_mix_fwd": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_int64, c_int, c_int, c_void_p]),
    "passion_region_mix_bwd": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "passion_augment": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "passion_adamw_step": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                   c_double, c_double, c_double, c_int64, c_int, c_void_p]),
    "passion_adamw_hyper": (c_int, [c_double, c_double, c_double, c_double, c_int64, c_void_p]),
    "passion_adamw_step_dev": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double,
                                       c_int, c_void_p, c_void_p]),
    "passion_write_words": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "passion_set_step_words": (None, [c_void_p]),
    "passion_stream_copy": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "passion_set_option": (c_int, [c_char_p, c_int]),
    "passion_prof_enable": (None, [c_int]),
    "passion_prof_filter": (None, [c_double, c_double]),
    "passion_prof_read": (c_int, [POINTER(ProfRec), c_int]),
    "passion_prof_peek": (c_int, [POINTER(ProfRec), c_int]),
}

_lib = None


class PassionHipError(RuntimeError):
    pass


def load():
    """Load libpassion_hip.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PassionHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C passion_amd/csrc`.  passion_amd has no CPU/eager fallback.")
    import torch  # noqa: F401  -- torch's bundled libamdhip64 must be the HIP runtime this library binds to:
    #                  loading ours first would pull a second runtime from /opt/rocm (kernels then fail with hipErrorNoDevice)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.passion_abi_version() != ABI_VERSION:
        raise PassionHipError(f"{LIB_PATH} has ABI version {lib.passion_abi_version()}, this package binds version {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise PassionHipError(f"{what} failed with code {rc}")


HOST_EXT_PATH = os.path.join(_HERE, "lib", "_passion_host.so")
_host = None


def host():
    """The C++ host extension (csrc/host_ext.cpp: autograd nodes of the conv block and the InstanceNorm statistics), bound
    to the entry points of the library load() returned.  PASSION_HOST_EXT=0 selects the Python autograd.Functions in ops.py
    instead (same C ABI underneath; kept for A/B timing of the host path)."""
    global _host
    if _host is not None:
        return _host or None
    if os.environ.get("PASSION_HOST_EXT", "1") == "0":
        _host = False
        return None
    lib = load()
    if not os.path.exists(HOST_EXT_PATH):
        raise PassionHipError(f"{HOST_EXT_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C passion_amd/csrc`")
    import importlib.util
    spec = importlib.util.spec_from_file_location("_passion_host", HOST_EXT_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not hasattr(mod, "abi_version") or mod.abi_version() != ABI_VERSION:
        raise PassionHipError(f"{HOST_EXT_PATH} was built against another descriptor layout than ABI version {ABI_VERSION}: rebuild it")
    addr = lambda name: ctypes.cast(getattr(lib, name), c_void_p).value
    mod.bind(addr("passion_conv3d_fwd"), addr("passion_conv3d_bwd"), addr("passion_conv_bwd_scratch_floats"),
             addr("passion_instnorm_stats"), addr("passion_instnorm_scratch_doubles"), addr("passion_conv3d_fwd_stats"),
             addr("passion_conv3d_fwd_grouped"), addr("passion_conv3d_bwd_grouped"))
    _host = mod
    return mod


def dtype_code(t):
    import torch
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise PassionHipError(f"unsupported activation dtype {t}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PassionHipError("passion_amd kernels need GPU tensors (no CPU fallback); got a "
                                  f"{t.device} tensor")


_raw_stream = _cur_dev = None


def stream_ptr():
    """raw hipStream_t of torch's current stream.  (torch.cuda.current_stream() builds a Stream object through several
    Python layers: 10 us a call, 250 calls a step -- the step is host-bound, so that was 10 % of it.)"""
    global _raw_stream, _cur_dev
    if _raw_stream is None:
        import torch
        _raw_stream, _cur_dev = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
    return c_void_p(_raw_stream(_cur_dev()))
