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
E4M3, E5M2 = 0, 1          # fp8 formats (CVCS_E4M3 / CVCS_E5M2)
ABI_VERSION = 14


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
        ("aniso", C.c_int32), ("stride_w", C.c_int32), ("pad_w", C.c_int32),
        ("in_row_pitch", C.c_int64), ("in_img_pitch", C.c_int64),
        ("res", C.c_void_p), ("res_ld", C.c_int64), ("res_scale", C.c_void_p), ("res_shift", C.c_void_p),
        ("in2", C.c_void_p), ("in2_ld", C.c_int64), ("Cin2", C.c_int32),
        ("mask", C.c_void_p), ("mask_ld", C.c_int64),
        ("res2", C.c_void_p), ("res2_ld", C.c_int64), ("res2_half", C.c_int32),
        ("mask_bits_out", C.c_void_p), ("mask_bits", C.c_void_p),
        ("in_up2", C.c_int32),
    ]


class Conv8Desc(C.Structure):
    """cvcs_conv8_desc (include/cvcs_hip.h): the fp8 3x3 convolution"""
    _fields_ = [
        ("in_", C.c_void_p), ("in_ld", C.c_int64), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("in_fmt", C.c_int32),
        ("wt", C.c_void_p),
        ("out", C.c_void_p), ("out_ld", C.c_int64), ("Cout", C.c_int32),
        ("scale_in", C.c_void_p), ("scale_w", C.c_void_p),
        ("relu", C.c_int32),
        ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p),
        ("stat_sum", C.c_void_p), ("stat_m2", C.c_void_p), ("stat_cnt", C.c_void_p),
    ]


class PackItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("Cin_pad", C.c_int32),
                ("Cout_pad", C.c_int32)]


class GatherItem(C.Structure):
    """cvcs_gather_item (include/cvcs_hip.h): one matrix of the table-driven weight gather"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p),
                ("base", C.c_int64), ("rs0", C.c_int64), ("rs1", C.c_int64), ("rs2", C.c_int64), ("cs0", C.c_int64), ("cs1", C.c_int64), ("cs2", C.c_int64),
                ("R", C.c_int32), ("Cp", C.c_int32), ("Rv", C.c_int32), ("Cv", C.c_int32), ("rd1", C.c_int32), ("rd2", C.c_int32),
                ("cd1", C.c_int32), ("cd2", C.c_int32), ("f32_out", C.c_int32), ("rv2", C.c_int32), ("cv2", C.c_int32), ("pad_", C.c_int32)]


class TailBwdDesc(C.Structure):
    """cvcs_tail_bwd_desc (include/cvcs_hip.h)"""
    _fields_ = [("out", C.c_void_p), ("out_ld", C.c_int64),
                ("g", C.c_void_p * 3), ("g_ld", C.c_int64 * 3), ("g_half", C.c_int32 * 3), ("dtype", C.c_int32),
                ("dz", C.c_void_p), ("dz_ld", C.c_int64),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("y", C.c_void_p * 2), ("y_ld", C.c_int64 * 2), ("mean", C.c_void_p * 2), ("invstd", C.c_void_p * 2),
                ("part_dz", C.c_void_p), ("part_dzx", C.c_void_p * 2),
                ("pool_g", C.c_void_p * 2), ("pool_g_ld", C.c_int64 * 2), ("pool_idx", C.c_void_p)]


CALL_MAX_INT, CALL_MAX_FLT = 28, 8


class Call(C.Structure):
    """cvcs_call (include/cvcs_hip.h): one launch of a plan replayed from C"""
    _fields_ = [("fn", C.c_void_p), ("nint", C.c_int32), ("nflt", C.c_int32), ("i", C.c_int64 * CALL_MAX_INT), ("f", C.c_float * CALL_MAX_FLT)]


class WgradDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_ld", C.c_int64), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("dy", C.c_void_p), ("dy_ld", C.c_int64), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("dw", C.c_void_p), ("Cin_real", C.c_int32),
        ("workspace", C.c_void_p),
        ("dtype", C.c_int32),
        ("aniso", C.c_int32), ("stride_w", C.c_int32), ("pad_w", C.c_int32),
        ("x_row_pitch", C.c_int64), ("x_img_pitch", C.c_int64),
        ("dil", C.c_int32),
        ("dbias", C.c_void_p),
        ("x_up2", C.c_int32),
    ]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/cvcs_hip.h declares
SIGNATURES = {
    "cvcs_last_error": (C.c_char_p, []),
    "cvcs_abi_version": (_i, []),
    "cvcs_sizeof_conv_desc": (_i, []),
    "cvcs_sizeof_wgrad_desc": (_i, []),
    "cvcs_sizeof_conv8_desc": (_i, []),
    "cvcs_conv3x3_fp8": (_i, [C.POINTER(Conv8Desc), _vp]),
    "cvcs_quantize_fp8": (_i, [_vp, _i64, _i64, _i, _vp, _i64, _i, _vp, _i, _vp]),
    "cvcs_fp8_update_scales": (_i, [_vp, _i, _f, _vp]),
    "cvcs_im2col": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_im2col_stem": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_col2im": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_phase_shuffle": (_i, [_vp, _i64, _i64, _i, _i, _i, _i, _vp, _i64, _i, _i, _vp]),
    "cvcs_phase_unshuffle": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _i64, _i64, _i, _vp]),
    "cvcs_dwconv3x3": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i64, _i, _vp]),
    "cvcs_dwconv3x3_wgrad_rows": (_i, [_i64]),
    "cvcs_dwconv_rows": (_i, [_i64, _i, _i]),
    "cvcs_dwconv": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "cvcs_dwconv_dgrad": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i64, _i, _i, _i, _vp]),
    "cvcs_dwconv_wgrad_rows": (_i, [_i64, _i, _i, _i]),
    "cvcs_dwconv_wgrad": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "cvcs_se_scale": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _f, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_image_dot": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_bn_add": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i, _vp, _i64, _i, _vp]),
    "cvcs_hardsigmoid": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp, _i64, _i, _vp]),
    "cvcs_dwconv3x3_wgrad": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _i, _vp]),
    "cvcs_drop_path_scales": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "cvcs_scale_rows_add": (_i, [_vp, _i64, _vp, _vp, _i64, _i, _i64, _i, _vp, _i64, _i, _vp]),
    "cvcs_sr_attention_fwd": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _vp, _i, _vp]),
    "cvcs_sr_attention_bwd_workspace": (_i64, [_i, _i, _i, _i, _i]),
    "cvcs_sr_attention_bwd": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i, _vp]),
    "cvcs_planes_from_nhwc": (_i, [_vp, _i64, _i, _i64, _i, _i, _vp, _i, _vp]),
    "cvcs_nhwc_from_planes": (_i, [_vp, _i, _i64, _i, _vp, _i64, _i, _i, _vp]),
    "cvcs_deconv_pack": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "cvcs_deconv_unpack_grad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "cvcs_relu_bwd_sum_bn": (_i, [C.POINTER(TailBwdDesc), _vp]),
    "cvcs_scale_unless_one": (_i, [_vp, _i64, _vp, _vp]),
    "cvcs_scale_unless_one_bf16": (_i, [_vp, _i64, _vp, _vp]),
    "cvcs_sizeof_call": (_i, []),
    "cvcs_replay": (_i, [_vp, _i, _vp, C.POINTER(C.c_int)]),
    "cvcs_gather_weights": (_i, [_vp, _i, _i, _vp]),
    "cvcs_scatter_weight_grads": (_i, [_vp, _i, _vp]),
    "cvcs_dropout": (_i, [_vp, _i64, _i64, _i, _vp, _i64, _vp, _f, _i, _vp]),
    "cvcs_counter_add": (_i, [_vp, C.c_uint64, _vp]),
    "cvcs_bn_act_q8": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _i64, _i, _vp, _i, _i, _vp]),
    "cvcs_bn_bwd_apply_q8": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i64, _vp, _vp, _i64, _i, _vp, _i,
                                  _i, _vp]),
    "cvcs_upsample2x_fwd_q8": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _i, _vp, _i, _i, _vp]),
    "cvcs_bn_add_act_q8": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _vp, _i64, _vp, _i64, _i, _vp, _i, _i, _vp]),
    "cvcs_conv_stat_rows": (_i, [C.POINTER(ConvDesc)]),
    "cvcs_wgrad_slices": (_i, [_i] * 8),
    "cvcs_conv2d": (_i, [C.POINTER(ConvDesc), _vp]),
    "cvcs_conv2d_wgrad": (_i, [C.POINTER(WgradDesc), _vp]),
    "cvcs_wgrad_workspace_floats": (_i64, [C.POINTER(WgradDesc)]),
    "cvcs_wgrad_takes_bias": (_i, [C.POINTER(WgradDesc)]),
    "cvcs_bn_add_act": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _vp, _i64, _i, _vp]),
    "cvcs_relu_bwd_sum": (_i, [_vp, _i64, _vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_maxpool3x3s2_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i64, _vp, _i, _vp]),
    "cvcs_maxpool3x3s2_bwd": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_dilate2x": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_regrid": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_pack_input_stem": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "cvcs_pack_stem_weight": (_i, [_vp, _i, _vp, _i, _vp]),
    "cvcs_unpack_stem_wgrad": (_i, [_vp, _i, _vp, _vp]),
    "cvcs_resize_bilinear_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_resize_bilinear_bwd": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_resize_bilinear_nchw_fwd": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp]),
    "cvcs_resize_bilinear_nchw_bwd": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp]),
    "cvcs_image_sum": (_i, [_vp, _i64, _i, _i, _i, _f, _vp, _i64, _i, _vp]),
    "cvcs_image_broadcast": (_i, [_vp, _i64, _i, _i, _i, _f, _vp, _i64, _i, _vp]),
    "cvcs_linear_head_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "cvcs_linear_head_bwd_rows": (_i, [_i64]),
    "cvcs_gn_rows": (_i, [_i]),
    "cvcs_layernorm_rows": (_i, [_i64]),
    "cvcs_layernorm_fwd": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _f, _vp, _i64, _vp, _vp, _i, _vp]),
    "cvcs_layernorm_bwd": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp, _vp, _vp, _vp, _i64, _vp, _i, _vp]),
    "cvcs_gelu": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp, _i64, _i, _vp]),
    "cvcs_pack_patches": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "cvcs_patch_merge": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i64, _i, _i, _vp]),
    "cvcs_window_gather": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_window_reverse": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "cvcs_window_attention_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _i, _vp]),
    "cvcs_window_attention_bwd_workspace_floats": (_i64, [_i, _i, _i, _i]),
    "cvcs_window_attention_bwd": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _i, _vp]),
    "cvcs_adaptive_avg_pool": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i, _i, _vp]),
    "cvcs_resize_bilinear_any": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _i64, _i, _i, _i, _vp]),
    "cvcs_gn_stats": (_i, [_vp, _i64, _i, _i, _i, _vp, _i, _vp]),
    "cvcs_gn_finalize": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "cvcs_gn_act_fwd": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp, _i, _vp, _i64, _i, _vp]),
    "cvcs_gn_act_bwd_reduce": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "cvcs_gn_bwd_finalize": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cvcs_gn_act_bwd_apply": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "cvcs_linear_head_bwd": (_i, [_vp, _i64, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i64, _vp, _i, _vp]),
    "cvcs_bn_finalize_workspace_floats": (_i, [_i, _i]),
    "cvcs_bn_finalize": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cvcs_bn_moments": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "cvcs_bn_finalize_moments": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "cvcs_bn_bwd_coeffs": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "cvcs_bn_act": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _i64, _i, _vp]),
    "cvcs_bn_bwd_rows": (_i, [_i64]),
    "cvcs_bn_bwd_chunk_lanes": (_i, [_i, _i]),
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
    "cvcs_gram_workspace_floats": (_i64, [_i64, _i]),
    "cvcs_gram": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _vp, _vp]),
    "cvcs_bn_gram_finalize": (_i, [_vp, _vp, _vp, _i, _i, _i64, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cvcs_bn_gram_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cvcs_bn_gram_mmat_workspace_floats": (_i64, [_i, _i]),
    "cvcs_bn_gram_mmat": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "cvcs_bn_gram_fold": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "cvcs_head_ce_rows": (_i, [_i64]),
    "cvcs_head_ce": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _f, _vp, _i64, _vp, _vp, _vp, _i, _vp]),
    "cvcs_sgd_step": (_i, [_vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _vp]),
    "cvcs_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _i, _vp]),
}

_lib = None
_recording = None          # the Recording that is capturing launches right now (None: plain eager calls)
_QUERIES = {"cvcs_last_error", "cvcs_abi_version", "cvcs_sizeof_conv_desc", "cvcs_sizeof_wgrad_desc", "cvcs_sizeof_conv8_desc", "cvcs_conv_stat_rows",
            "cvcs_wgrad_slices", "cvcs_wgrad_workspace_floats", "cvcs_wgrad_takes_bias", "cvcs_bn_finalize_workspace_floats", "cvcs_bn_bwd_rows",
            "cvcs_head_bwd_rows", "cvcs_ce_workspace_floats", "cvcs_linear_head_bwd_rows", "cvcs_gn_rows", "cvcs_layernorm_rows",
            "cvcs_window_attention_bwd_workspace_floats", "cvcs_dwconv3x3_wgrad_rows", "cvcs_dwconv_rows", "cvcs_dwconv_wgrad_rows", "cvcs_sr_attention_bwd_workspace", "cvcs_sizeof_call", "cvcs_replay",
            "cvcs_gram_workspace_floats", "cvcs_bn_gram_mmat_workspace_floats", "cvcs_head_ce_rows", "cvcs_bn_bwd_chunk_lanes"}
C_REPLAY = os.environ.get("CVCS_C_REPLAY", "1") == "1"     # single-stream replays without timers run from C (cvcs_replay)
pending_tag = None         # (kernel family, algorithmic flops) of the NEXT launch, set by ops.conv2d / ops.conv2d_wgrad


lane = 0                   # 0: the launch belongs to the main chain of the plan; 1: it may run beside it (ops.side_lane())


class Recording:
    """A launch plan: the sequence of C-ABI launches (function, arguments without the stream) that one pass of a network
    issues for one input shape, captured while it runs eagerly ONCE and replayed every later step with no Python between
    the launches but this loop.  Every pointer in it is a persistent engine buffer; descriptors are kept alive by the
    argument tuples.  ("HIP streams and graphs instead of a tracing compiler": this is the host-side launch list; the same
    replay runs under a HIP-graph capture unchanged.)  Host callbacks (data-parallel bucket hooks) are recorded in place.

    Two lanes: launches recorded inside `side_lane()` (weight gradients: nothing of the backward chain waits for them) are
    replayed on a second HIP stream when `replay` is given one - ordered after everything recorded before them (an event
    of the main stream), with explicit `wait_side(i)` markers where the main chain is about to overwrite a buffer side
    launch i reads, and a full join in front of every host callback and at the end."""

    def __init__(self):
        self.items = []     # (fn, args, tag, lane) | (None, callable, None, 0) | ("wait", index | None, None, 0)
        self._ev = None     # per-item events of the two-lane replay (created once)
        self._c = None      # the plan as cvcs_call arrays between host callbacks (built on the first C replay)

    def __enter__(self):
        global _recording
        assert _recording is None, "recordings do not nest"
        lib()               # make sure the library is loaded before the proxy is consulted
        _recording = self
        return self

    def __exit__(self, *exc):
        global _recording
        _recording = None
        return False

    def host(self, fn):
        """run fn() now and at this point of every replay"""
        self.items.append((None, fn, None, 0))
        fn()

    def last_index(self) -> int:
        return len(self.items) - 1

    def wait_side(self, index=None):
        """the main chain must not pass this point before side launch `index` (None: every side launch so far) is done"""
        self.items.append(("wait", index, None, 0))

    def _compile(self):
        """segments of the single-stream plan: ("c", cvcs_call array, n, first item index) | ("host", callable)"""
        segs, cur, first = [], [], 0

        def flush():
            if cur:
                arr = (Call * len(cur))()
                for c, (fn, args) in zip(arr, cur):
                    types = fn.argtypes[:-1]
                    assert len(types) == len(args), fn.__name__
                    c.fn = C.cast(fn, C.c_void_p).value
                    ni = nf = 0
                    for a, t in zip(args, types):
                        if t is C.c_float:
                            c.f[nf] = float(a)
                            nf += 1
                            continue
                        if a is None:
                            v = 0
                        elif isinstance(a, int):
                            v = a
                        elif hasattr(a, "_obj"):                      # byref(descriptor)
                            v = C.addressof(a._obj)
                        elif isinstance(a, (C.Array, C.Structure)):
                            v = C.addressof(a)
                        elif isinstance(a, C._SimpleCData):
                            # (a mutable ctypes scalar would be snapshotted here but re-read by the Python replay: no wrapper passes one)
                            raise TypeError(f"{fn.__name__}: ctypes scalar objects cannot be recorded in a plan (pass a plain int / float)")
                        else:
                            v = C.cast(a, C.c_void_p).value or 0
                        c.i[ni] = v if v < (1 << 63) else v - (1 << 64)
                        ni += 1
                    assert ni < CALL_MAX_INT and nf <= CALL_MAX_FLT, fn.__name__
                    c.nint, c.nflt = ni, nf
                segs.append(("c", arr, len(cur), first))
                cur.clear()

        for idx, (fn, args, tag, _) in enumerate(self.items):
            if fn == "wait":
                continue
            if fn is None:
                flush()
                segs.append(("host", args, 0, idx))
                continue
            if not cur:
                first = idx
            cur.append((fn, args))
        flush()
        self._c = segs

    def _replay_c(self, stream: int):
        h = _load()
        if self._c is None:
            self._compile()
        failed = C.c_int(-1)
        for kind, obj, n, first in self._c:
            if kind == "host":
                obj()
                continue
            rc = h.cvcs_replay(obj, n, stream, C.byref(failed))
            if rc != 0:
                launches = [it for it in self.items[first:] if it[0] is not None and it[0] != "wait"]
                name = launches[failed.value][0].__name__ if 0 <= failed.value < len(launches) else "cvcs_replay"
                check(rc, name)

    def replay(self, stream: int, timers=None, side=None):
        """stream: raw handle of the main stream.  side: a torch.cuda.Stream for the side lane (None: one stream, in order)"""
        if side is None and timers is None and C_REPLAY:
            return self._replay_c(stream)
        if side is None:
            for fn, args, tag, _ in self.items:
                if fn is None:
                    args()
                    continue
                if fn == "wait":
                    continue
                if timers is not None and tag is not None:
                    ev = timers.bracket(*tag)
                    ev[0].record()
                    rc = fn(*args, stream)
                    ev[1].record()
                else:
                    rc = fn(*args, stream)
                if rc != 0:
                    check(rc, fn.__name__)
            return
        import torch
        main = torch.cuda.current_stream()
        assert main.cuda_stream == stream
        if self._ev is None:
            self._ev = {i: (torch.cuda.Event(), torch.cuda.Event()) for i, it in enumerate(self.items) if it[3] == 1}
            self._join = torch.cuda.Event()
        side_h = side.cuda_stream
        pending = False     # side work issued since the last full join
        for i, (fn, args, tag, ln) in enumerate(self.items):
            if fn is None or fn == "wait":
                if fn == "wait" and args is not None:
                    main.wait_event(self._ev[args][1])
                elif pending:
                    self._join.record(side)
                    main.wait_event(self._join)
                    pending = False
                if fn is None:
                    args()
                continue
            if ln == 1:
                fork, done = self._ev[i]
                fork.record(main)
                side.wait_event(fork)
                if timers is not None and tag is not None:
                    ev = timers.bracket(*tag)
                    ev[0].record(side)
                    rc = fn(*args, side_h)
                    ev[1].record(side)
                else:
                    rc = fn(*args, side_h)
                done.record(side)
                pending = True
            elif timers is not None and tag is not None:
                ev = timers.bracket(*tag)
                ev[0].record()
                rc = fn(*args, stream)
                ev[1].record()
            else:
                rc = fn(*args, stream)
            if rc != 0:
                check(rc, fn.__name__)
        if pending:
            self._join.record(side)
            main.wait_event(self._join)


class side_lane:
    """launches issued inside this context may run beside the main chain of a recorded plan (Recording)"""

    def __enter__(self):
        global lane
        self._prev, lane = lane, 1

    def __exit__(self, *exc):
        global lane
        lane = self._prev
        return False


class _Proxy:
    def __init__(self, h):
        self._h = h

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if name in _QUERIES:
            return fn

        def call(*args):
            global pending_tag
            tag, pending_tag = pending_tag, None
            _recording.items.append((fn, args[:-1], tag, lane))
            return fn(*args)
        return call


_proxy = None


def lib():
    """Load (once) and return the ctypes handle; raises CvcsError if the HIP library is not built.  While a Recording is
    open the handle is a proxy that also appends every launch to it."""
    if _recording is not None and _lib is not None:
        return _proxy
    return _load()


def _load():
    global _lib, _proxy
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
        if h.cvcs_abi_version() != ABI_VERSION:
            raise CvcsError("libcvcs_hip.so ABI version mismatch")
        if (h.cvcs_sizeof_conv_desc() != C.sizeof(ConvDesc) or h.cvcs_sizeof_wgrad_desc() != C.sizeof(WgradDesc) or
                h.cvcs_sizeof_conv8_desc() != C.sizeof(Conv8Desc)):
            raise CvcsError("descriptor layout of cvcs_amd/_lib.py differs from the one libcvcs_hip.so was compiled with")
        _check_replay_contract()
        _lib = h
        _proxy = _Proxy(h)
    return _lib


def _check_replay_contract():
    """cvcs_replay (csrc/api.hip) calls every launch entry point through ONE prototype - 28 integer-class slots, then 8 floats - which is only
    right on the x86-64 System V ABI and only for entry points whose arguments are pointers / int / int64 / uint64 (general registers, then the
    stack, in order), at most 8 floats (xmm0-7), no double, no struct by value, the stream last.  Enforced here for every entry a plan can record;
    on another machine the plans are replayed from Python."""
    global C_REPLAY
    import platform
    if platform.machine() not in ("x86_64", "AMD64"):
        C_REPLAY = False
        return
    ints = (C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_char_p)
    for name, (res, args) in SIGNATURES.items():
        if name in _QUERIES:
            continue
        assert res is C.c_int and args and args[-1] is C.c_void_p, f"{name}: a launch entry point returns int and takes the stream last"
        nf = sum(1 for t in args if t is C.c_float)
        ni = len(args) - 1 - nf
        for t in args[:-1]:
            assert t is C.c_float or t in ints or (isinstance(t, type) and issubclass(t, C._Pointer)), f"{name}: argument type {t} cannot go through cvcs_replay"
        assert nf <= CALL_MAX_FLT and ni < CALL_MAX_INT, f"{name}: {ni} integer-class / {nf} float arguments exceed the cvcs_call record"


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = _load().cvcs_last_error().decode(errors="replace")
        raise CvcsError(f"{what or 'cvcs'} failed ({rc}): {msg}")
