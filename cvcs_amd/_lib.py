"""ctypes binding of libcvcs_hip.so (the C-ABI declared in include/cvcs_hip.h).

The library is built in-tree by `make -C cvcs_amd/csrc` (or `__graft_entry__.build()`); there is no fallback:
if it is missing the product path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcvcs_hip.so")

F32, BF16 = 0, 1


class CvcsError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("in_ld", C.c_int64), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("wt", C.c_void_p), ("bias", C.c_void_p),
        ("out", C.c_void_p), ("out_ld", C.c_int64), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32),
        ("relu", C.c_int32), ("pixel_shuffle", C.c_int32),
        ("stat_sum", C.c_void_p), ("stat_m2", C.c_void_p), ("stat_cnt", C.c_void_p),
        ("dtype", C.c_int32),
        ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p), ("post_scale", C.c_void_p), ("post_shift", C.c_void_p),
        ("Cin_valid", C.c_int32),
        ("pool_out", C.c_void_p), ("pool_ld", C.c_int64),
        ("bwd_y", C.c_void_p), ("bwd_y_ld", C.c_int64),
        ("bwd_scale", C.c_void_p), ("bwd_shift", C.c_void_p), ("bwd_mean", C.c_void_p), ("bwd_invstd", C.c_void_p),
        ("bwd_mode", C.c_int32),
        ("bwd_part_dz", C.c_void_p), ("bwd_part_dzx", C.c_void_p),
    ]


class PackItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("Cin_pad", C.c_int32),
                ("reserved", C.c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_ld", C.c_int64), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("dy", C.c_void_p), ("dy_ld", C.c_int64), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("dw", C.c_void_p), ("Cin_real", C.c_int32),
        ("workspace", C.c_void_p),
        ("dtype", C.c_int32),
    ]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/cvcs_hip.h declares
SIGNATURES = {
    "cvcs_last_error": (C.c_char_p, []),
    "cvcs_abi_version": (_i, []),
    "cvcs_sizeof_conv_desc": (_i, []),
    "cvcs_sizeof_wgrad_desc": (_i, []),
    "cvcs_conv_stat_rows": (_i, [C.POINTER(ConvDesc)]),
    "cvcs_wgrad_slices": (_i, [_i] * 8),
    "cvcs_conv2d": (_i, [C.POINTER(ConvDesc), _vp]),
    "cvcs_conv2d_wgrad": (_i, [C.POINTER(WgradDesc), _vp]),
    "cvcs_bn_finalize_workspace_floats": (_i, [_i, _i]),
    "cvcs_bn_finalize": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cvcs_bn_moments": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "cvcs_bn_finalize_moments": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "cvcs_bn_bwd_coeffs": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "cvcs_bn_act": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _i64, _i, _vp]),
    "cvcs_bn_bwd_rows": (_i, [_i64]),
    "cvcs_bn_bwd_reduce": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "cvcs_bn_bwd_finalize": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cvcs_bn_bwd_apply": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i64,
                               _vp, _i, _vp]),
    "cvcs_colsum_finalize": (_i, [_vp, _i, _i, _vp, _vp]),
    "cvcs_colsum_partial": (_i, [_vp, _i64, _i64, _i, _vp, _i, _vp]),
    "cvcs_upsample2x_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_upsample2x_bwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_pack_input": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "cvcs_pack_conv_weight": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "cvcs_pack_conv_weights": (_i, [_vp, _i, _i, _vp]),
    "cvcs_pack_convT_weight": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "cvcs_head_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "cvcs_head_fold": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "cvcs_head_unfold_grad": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "cvcs_head_argmax": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "cvcs_label_stitch": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "cvcs_head_bwd_rows": (_i, [_i64]),
    "cvcs_head_bwd": (_i, [_vp, _i64, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i64, _vp, _i, _vp]),
    "cvcs_ce_workspace_floats": (_i, [_i64]),
    "cvcs_ce_weight_sum": (_i, [_vp, _i, _i, _i, _i64, _vp, _i, _vp, _vp]),
    "cvcs_ce_fwd_bwd": (_i, [_vp, _vp, _i, _i, _i, _i64, _vp, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "cvcs_argmax_confusion": (_i, [_vp, _i, _i, _i64, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "cvcs_label_confusion": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "cvcs_vote_labels": (_i, [_vp, _i, _i64, _vp, _vp]),
    "cvcs_crop_tiles": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "cvcs_argmax_stitch": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "cvcs_gather_tiles": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "cvcs_label_histogram": (_i, [_vp, _i64, _i, _vp, _vp]),
    "cvcs_sgd_step": (_i, [_vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _vp]),
    "cvcs_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _i, _vp]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises CvcsError if the HIP library is not built."""
    global _lib
    if _lib is None:
        # torch first: its wheel bundles its own libamdhip64, and the HIP runtime that is loaded FIRST is the one this library
        # binds to - loaded before torch, the system runtime and torch's would coexist in the process and the launches of
        # this library would not see torch's device ("no ROCm-capable device is detected")
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise CvcsError(f"{LIB_PATH} not found: build it with `make -C cvcs_amd/csrc` "
                            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.cvcs_abi_version() != 3:
            raise CvcsError("libcvcs_hip.so ABI version mismatch")
        if h.cvcs_sizeof_conv_desc() != C.sizeof(ConvDesc) or h.cvcs_sizeof_wgrad_desc() != C.sizeof(WgradDesc):
            raise CvcsError("descriptor layout of cvcs_amd/_lib.py differs from the one libcvcs_hip.so was compiled with")
        _lib = h
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().cvcs_last_error().decode(errors="replace")
        raise CvcsError(f"{what or 'cvcs'} failed ({rc}): {msg}")
