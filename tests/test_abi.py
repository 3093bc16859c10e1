"""CPU: the C-ABI library builds/loads and exports every symbol include/passion_hip.h declares
(no compute calls without a GPU); the ctypes table in passion_amd/_lib.py covers the same set."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "passion_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(passion_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_surface():
    syms = header_symbols()
    for s in ("passion_conv3d_fwd", "passion_conv3d_bwd", "passion_instnorm_stats", "passion_upsample_fwd",
              "passion_wce_dice_fwd", "passion_kl_fwd", "passion_proto_fwd", "passion_attn_scores_fwd"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from passion_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in passion_hip.h but not exported"
    assert lib.passion_abi_version() == 2


def test_ctypes_table_matches_header():
    from passion_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_struct_layout_matches_header():
    from passion_amd import _lib
    # passion_seg_t: 4 pointers + int64 + 4 int32 + the dx_add / dx_add2 pointers (round 6) = 72 bytes; conv desc: 12 x 4-byte fields + 4 segments
    assert ctypes.sizeof(_lib.Seg) == 72
    assert ctypes.sizeof(_lib.ConvDesc) == 48 + 4 * 72
    assert _lib.Seg.dx_add.offset == 56 and _lib.Seg.dx_add2.offset == 64
    assert ctypes.sizeof(_lib.ProfRec) == 24 + 64      # {kind, ms, bytes, flops, char name[64]}


def test_product_path_has_no_cpu_fallback():
    import torch
    from passion_amd import _lib, ops
    with pytest.raises(_lib.PassionHipError):
        ops.instnorm_stats(torch.zeros(1, 2, 2, 2, 8))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "passion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("# oracle", ""), f"{f} mentions the oracle"


def test_host_extension_loads_and_binds():
    """the C++ autograd nodes (csrc/host_ext.cpp) import on a CPU-only box and get the kernel entry points from ctypes"""
    from passion_amd import _lib
    h = _lib.host()
    assert h is not None
    for name in ("bind", "conv_block", "instnorm_stats"):
        assert callable(getattr(h, name))
