"""ctypes binding of libdeeplip_hip.so (C ABI: include/deeplip_hip.h).

There is no CPU fallback anywhere in this package: if the library is missing or fails to load,
``lib()`` raises and every op that needs it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # must be imported first: the library binds to the HIP runtime torch already loaded

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DLIP_LIB_PATH") or os.path.join(_PKG, "lib", "libdeeplip_hip.so")  # env override: A/B builds
ABI_VERSION = 49
LIFT_WORDS = 4098
LIFT_BCAST = 2048

_lock = threading.Lock()
_lib = None
_status = None
_status_np = None

c_f = C.c_void_p      # device float* (passed as integer address)
c_i32 = C.c_int32
c_i64 = C.c_int64
c_stream = C.c_void_p


class ConvDesc(C.Structure):
    """struct dlip_conv_desc"""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "C", "K", "R", "S", "stride_h", "stride_w", "pad_h", "pad_w", "dil_h", "dil_w",
        "Ho", "Wo", "ldx", "ldy", "ldr")]


# name -> argtypes (restype is int for all but dlip_error_string); mirrors include/deeplip_hip.h
SIGNATURES = {
    "dlip_abi_version": [],
    "dlip_conv_nhwc_f32": [C.POINTER(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_stream],
    "dlip_conv_nhwc_f16x3": [C.POINTER(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_stream],
    "dlip_conv2_nhwc_f16x3": [C.POINTER(ConvDesc), c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f,
                              c_i32, c_stream],
    "dlip_conv_pool_partial_bytes": [C.POINTER(ConvDesc), C.POINTER(C.c_int32)],
    "dlip_conv_pool_f16x3": [C.POINTER(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_f, c_i32, c_i32, c_stream],
    "dlip_pool_finish_f32": [c_f, c_i64, c_i32, c_i32, c_i32, c_f, c_i32, c_i32, c_i32, c_i32, c_f, c_stream],
    "dlip_set_status_words": [c_f],
    "dlip_status_scope": [c_f],
    "dlip_span_scope_begin": [c_f, c_f, c_i32],
    "dlip_span_scope_end": [c_stream, C.POINTER(C.c_int32)],
    "dlip_range_scope_begin": [c_f, c_i32],
    "dlip_range_scope_end": [c_stream],
    "dlip_debug_set": [c_i32, c_i32],
    "dlip_bn_rows_chunks": [c_i32],
    "dlip_bn_rows_train_fwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, C.c_float, C.c_float, C.c_float, c_i32, c_i32, c_f, c_stream],
    "dlip_conv_stats_chunks": [C.POINTER(ConvDesc)],
    "dlip_conv_nhwc_stats_f16x3": [C.POINTER(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_stream],
    "dlip_bn_rows_train_bwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, C.c_float, c_i32, c_f, c_stream],
    "dlip_bn_prelu_rows_train_fwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, C.c_float, C.c_float, c_f, c_stream],
    "dlip_bn_prelu_maxpool_train_fwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_i32, c_i32, C.c_float, C.c_float, c_f, c_stream],
    "dlip_bn_prelu_maxpool_train_bwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_f, c_stream],
    "dlip_bn_add_prelu_rows_train_fwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, C.c_float, C.c_float, c_f, c_stream],
    "dlip_bn_add_prelu_rows_train_bwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_f, c_stream],
    "dlip_wgrad_operand_split_bn_f32": [c_f, c_f, c_i64, c_i64, c_i32, c_f, c_f, c_f, c_f, c_f, C.c_float, c_f, c_stream],
    "dlip_wgrad_chwn_bn_f32": [c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_i32, c_f, c_f, c_f, c_f, c_f, C.c_float, c_f, c_stream],
    "dlip_meanstd_pool_bn_f32": [c_f, c_f, c_f, c_f, c_f, C.c_float, c_f, c_i32, c_i32, c_i32, c_stream],
    "dlip_meanstd_pool_bwd_bn_f32": [c_f, c_f, c_f, c_f, c_f, C.c_float, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_stream],
    "dlip_meanstd_bwd_coef_f32": [c_f, c_f, c_f, c_i32, c_i32, c_i32, c_stream],
    "dlip_bn_rows_train_bwd_ms_f32": [c_f, c_i32, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, C.c_float, c_f, c_stream],
    "dlip_bn_rows_train_bwd_sums_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, C.c_float, c_i32, c_f, c_f, c_i32,
                                        c_stream],
    "dlip_wgrad_operand_split_bnbwd_f32": [c_f, c_f, c_f, c_i64, c_i64, c_i32, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, C.c_float, c_i32, c_f, c_f,
                                           c_i32, c_f, c_i32, c_stream],
    "dlip_wgrad_chwn_bnbwd_f32": [c_f, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_i32, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, C.c_float, c_i32, c_f,
                                  c_f, c_stream],
    "dlip_bn_apply_rows_f32": [c_f, c_f, c_f, c_f, c_f, c_f, C.c_float, c_f, c_i32, c_i32, c_stream],
    "dlip_dropout_keep_f32": [c_f, c_f, c_f, c_i64, C.c_float, C.c_float, c_stream],
    "dlip_chomp_concat_f32": [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), c_i32, c_f, c_i32, c_i32, c_i32, c_stream],
    "dlip_upsample_zero_split_f32": [c_f, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_bn_prelu_rows_train_bwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_f, c_stream],
    "dlip_colsum_rows_f32": [c_f, c_f, c_f, c_i32, c_i32, c_stream],
    "dlip_meanstd_pool_bwd_f32": [c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_stream],
    "dlip_permute3_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_pow2_lift_f32": [c_f, c_f, c_i64, C.c_float, c_stream],
    "dlip_pow2_scale_f32": [c_f, c_f, c_i64, C.c_float, c_stream],
    "dlip_split_pack_scaled_pad_f32": [c_f, c_f, c_f, c_i64, c_i32, c_i32, c_stream],
    "dlip_split_pack_scaled_f32": [c_f, c_f, c_f, c_i64, c_i32, c_stream],
    "dlip_split_weights_rows_f32": [c_f, c_f, c_f, c_i32, c_i32, c_stream],
    "dlip_split_weights_perm_f32": [c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_fill_from_scalar_f32": [c_f, c_f, c_i32, c_stream],
    "dlip_aam_margin_f32": [c_f, c_f, c_f, c_f, c_i32, c_i32, C.c_float, c_i32, c_i32, c_stream],
    "dlip_split_weights_multi_f32": [c_f, c_f, c_i32, c_i32, c_stream],
    "dlip_wgrad_operand_split_f32": [c_f, c_f, c_i64, c_i64, c_i32, c_f, c_f, c_stream],
    "dlip_wgrad_operand_f32": [c_f, c_f, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f, c_stream],
    "dlip_stem_wgrad_chwn_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_wgrad_conv_f16x3": [c_f, c_f, c_f, c_f, c_f, c_f] + [c_i32] * 15 + [c_stream],
    "dlip_wgrad_chwn_f32": [c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_f, c_i32, c_f, c_stream],
    "dlip_tap_gather_f32": [c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_upsample_zero_f32": [c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_prelu_rows_fwd_f32": [c_f, c_f, c_f, c_i64, c_i32, c_stream],
    "dlip_add_prelu_rows_fwd_f32": [c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_stream],
    "dlip_prelu_rows_bwd_f32": [c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_stream],
    "dlip_maxpool3x3s2_bwd_f32": [c_f, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_stream],
    "dlip_maxpool3x3s2_idx_f32": [c_f, c_f, C.c_void_p, c_i64, c_i32, c_i32, c_i32, c_stream],
    "dlip_maxpool3x3s2_bwd_idx_f32": [C.c_void_p, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_stream],
    "dlip_row_broadcast_f32": [c_f, c_f, c_f, c_i64, c_i32, c_i32, C.c_float, c_stream],
    "dlip_stem_im2col_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_stem_wgrad_operand_f32": [c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_split_stem_weights_f32": [c_f, c_f, c_f, c_i32, c_stream],
    "dlip_mul_mask_f32": [c_f, c_f, c_f, c_i64, C.c_float, c_stream],
    "dlip_conv_workspace_bytes": [],
    "dlip_conv_set_workspace": [c_f, c_i64, c_stream],
    "dlip_split_pack_f32": [c_f, c_f, C.c_int64, c_i32, c_stream],
    "dlip_split_unpack_f32": [c_f, c_f, C.c_int64, c_i32, c_stream],
    "dlip_conv_plan": [C.POINTER(ConvDesc), c_i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "dlip_conv_kernel_kind": [C.POINTER(ConvDesc)],
    "dlip_plan_begin": [c_stream],
    "dlip_plan_end": [c_stream, C.POINTER(C.c_void_p)],
    "dlip_plan_run": [C.c_void_p, c_stream],
    "dlip_plan_launches": [C.c_void_p],
    "dlip_plan_destroy": [C.c_void_p],
    "dlip_stem3d_bn_act_f32": [c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_stem3d_bn_act_f16x3": [c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_stem3d_pool_f16x3": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_stem3d_pool_u8_f16x3": [c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_stem3d_pool_workspace_bytes": [c_i32, c_i32, c_i32, c_i32],
    "dlip_selftest_lds_oob": [c_f, c_i32, c_stream],
    "dlip_maxpool3x3s2_nhwc_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_avgpool_nhwc_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_stream],
    "dlip_time_mean_f32": [c_f, c_f, c_i32, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_group_mean_f32": [c_f, c_f, c_f, c_i32, c_i32, c_stream],
    "dlip_mask_frames_f32": [c_f, c_f, c_f, c_i32, c_i32, c_i32, c_stream],
    "dlip_meanstd_pool_f32": [c_f, c_f, c_i32, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_attentive_stat_pool_f32": [c_f, c_f, c_f, c_f, c_f, c_i32, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_attentive_stat_pool_bwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_nct_to_ntc_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_ntc_to_nct_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_stream],
    "dlip_nct_to_ntc_split_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_ingest_rgb_u8": [c_f, c_f, c_i64, c_i32, c_i32, c_stream],
    "dlip_affine_act_f32": [c_f, c_f, c_f, c_f, c_i64, c_i32, C.c_float, c_i32, c_stream],
    "dlip_znorm_cat_f32": [c_f, c_i32, c_f, c_i32, c_f, c_i32, c_i32, c_stream],
    "dlip_znorm_cat_pooled_f32": [c_f, c_i32, c_f, c_i64, c_i32, c_i32, c_i32, c_f, c_i32, c_i32, c_f, c_i32, c_i32, c_stream],
    "dlip_l2_normalize_f32": [c_f, c_f, c_i32, c_i32, C.c_float, c_stream],
    "dlip_pair_cosine_f32": [c_f, c_i32, c_i32, c_f, c_f, c_f, c_i32, c_i32, C.c_float, C.c_float, c_i32, c_stream],
    "dlip_plda_llr_f32": [c_f, c_i32, c_i32, c_f, c_f, c_f, c_f, c_i32, c_stream],
    "dlip_logits_argmax_f32": [c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_margin_ce_loss_f32": [c_f, c_f, c_f, c_i32, c_i32, C.c_float, C.c_float, c_stream],
    "dlip_lowfer_cat_f32": [c_f, c_f, c_f, c_i32, c_i32, c_stream],
    "dlip_frame_preemph_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, C.c_float, c_stream],
    "dlip_powspec_f32": [c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_powspec_dft64_f32": [c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_powspec_wave_fft64_f32": [c_f, c_f, c_f, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, C.c_double, c_i32, c_i32, c_stream],
    "dlip_log_floor_f32": [c_f, c_f, c_i64, c_stream],
    "dlip_cmvn_nct_f32": [c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_delta_nct_f32": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_crop_normalize_u8": [c_f, c_f, c_f, c_i32, c_f, c_i64, c_i32, c_i32, c_i32, c_i32, c_stream],
    "dlip_bn1d_train_fwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, C.c_float, C.c_float, C.c_float, c_stream],
    "dlip_bn1d_train_bwd_f32": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_stream],
    "dlip_lrelu_bwd_f32": [c_f, c_f, c_f, c_i64, C.c_float, c_stream],
    "dlip_l1_sum_f32": [c_f, c_f, c_i64, c_stream],
    "dlip_l1_sign_f32": [c_f, c_f, c_f, C.c_float, c_i64, c_stream],
    "dlip_colsum_f32": [c_f, c_f, c_i32, c_i32, c_stream],
    "dlip_margin_ce_bwd_f32": [c_f, c_f, c_f, c_i32, c_i32, C.c_float, C.c_float, C.c_float, c_f, c_stream],
    "dlip_l2_normalize_bwd_f32": [c_f, c_f, c_f, c_i32, c_i32, C.c_float, c_stream],
    "dlip_gemm_small_f32": [c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_stream],
}


class DeepLipHipError(RuntimeError):
    pass


class DeepLipRangeError(DeepLipHipError):
    """An activation left the range of the split-fp16 ("f16x3") arithmetic (|v| >= 65520): the results of the
    launches since the last check are invalid.  Recourse: pack that model in the exact mode,
    ``deeplip_amd.packing.set_precision("f32")`` (same engine, fp32 MFMA)."""


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raises if it is missing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DeepLipHipError(
                f"{LIB_PATH} not found: build it with `python -m deeplip_amd.build` "
                "(or __graft_entry__.build()). deeplip_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the ABI lost a symbol
            fn.argtypes = args
            fn.restype = C.c_int
        l.dlip_conv_workspace_bytes.restype = C.c_int64
        l.dlip_conv_pool_partial_bytes.restype = C.c_int64
        l.dlip_stem3d_pool_workspace_bytes.restype = C.c_int64
        l.dlip_source_sha.argtypes = []
        l.dlip_source_sha.restype = C.c_char_p
        l.dlip_error_string.argtypes = [C.c_int]
        l.dlip_error_string.restype = C.c_char_p
        v = l.dlip_abi_version()
        if v != ABI_VERSION:
            raise DeepLipHipError(f"libdeeplip_hip.so ABI {v} != binding ABI {ABI_VERSION}; rebuild")
        _lib = l
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().dlip_error_string(code).decode()
        raise DeepLipHipError(f"{what} failed: {msg} (code {code})")


def ptr(t) -> int | None:
    """Device address of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_handle() -> int:
    if _status is None:
        status_words()      # first launch of the process: register the range-status block
    return torch.cuda.current_stream().cuda_stream


# Workspace of the conv kernel's balanced work split: one torch-owned block per (device, stream),
# registered with the library the first time that stream launches a convolution, so the library itself
# never allocates device memory.
_workspaces: dict = {}


def ensure_conv_workspace() -> None:
    st = torch.cuda.current_stream()
    key = (torch.cuda.current_device(), st.cuda_stream)
    if key in _workspaces:
        return
    nbytes = int(lib().dlip_conv_workspace_bytes())
    if nbytes <= 0:
        raise DeepLipHipError("dlip_conv_workspace_bytes failed (no ROCm device?)")
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    check(lib().dlip_conv_set_workspace(buf.data_ptr(), nbytes, st.cuda_stream), "dlip_conv_set_workspace")
    _workspaces[key] = buf


# ---- diagnostic overrides (tests, tools): dlip_debug_set ----
DBG_CONV_TILE, DBG_DMA_TILE, DBG_DMA_ENABLE, DBG_STREAMK, DBG_WIN, DBG_NINNER, DBG_ROWS, DBG_ROWS2D, DBG_BN_FUSED, DBG_ROWS_TAIL = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9


DEBUG = {}          # what this process set through debug_set (key -> value; -1 / absent = the built-in choice)


def debug_set(key: int, value: int = -1) -> None:
    check(lib().dlip_debug_set(key, value), "dlip_debug_set")
    DEBUG[key] = value


# ---- range status of the f16x3 arithmetic: eight words in host-pinned, device-visible memory ----
_STATUS_NAMES = ("a convolution / linear epilogue", "the fused stem + pool", "split_pack (a model input or gradient operand)",
                 "statistics pooling")
_ST_LOW = 4


def status_words():
    """Registers (once) the status block the kernels report range violations to.  It lives in pinned host memory,
    which the GPU addresses directly: the host can look at it at any time without synchronising."""
    global _status, _status_np
    if _status is None:
        t = torch.zeros(8, dtype=torch.int32).pin_memory()
        check(lib().dlip_set_status_words(t.data_ptr()), "dlip_set_status_words")
        _status_np = t.numpy()      # same memory: a forward's check is one numpy read
        _status = t
    return _status


def _range_error(words) -> "DeepLipRangeError":
    high = [n for n, v in zip(_STATUS_NAMES, words[:4]) if v]
    recourse = ("results since the last check are invalid. Run that model in the exact mode -- arith 'f32' (deeplip_amd.arith / "
                "--arith / model.arith / DLIP_ARITH), i.e. deeplip_amd.packing.set_precision('f32'): exact fp32 MFMA, same engine; "
                "arith 'auto' does it by itself for the batch concerned.")
    if high:
        return DeepLipRangeError("f16x3 arithmetic: an activation with |v| >= 65520 (not representable as hi + lo fp16) was "
                                 f"produced by {', '.join(high)}; " + recourse)
    fam = words[_ST_LOW] - 1
    who = _STATUS_NAMES[fam] if 0 <= fam < len(_STATUS_NAMES) else "a split-format producer"
    return DeepLipRangeError(f"f16x3 arithmetic: {who} produced a tensor whose largest magnitude is below 2^-2 = 0.25: its "
                             "lo halves are fp16 subnormals and the result is no longer fp32-grade (relative error 3e-8 / max|v|); "
                             + recourse)


class StatusBlock:
    """A status block of its own (dlip_status_scope): int32[8] in pinned host memory, handed to the launches a thread makes inside
    ``scope()`` -- a recorded step plan keeps reporting to it on every replay, so the host can tell WHICH plan's batch left the
    range.  Live blocks are also looked at by ``check_range()``: nothing reported anywhere goes unseen."""

    def __init__(self):
        self.t = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.np = self.t.numpy()
        self.private = False        # True: the owner settles every report itself (a pipeline's per-batch f32 re-run); check_range() skips it
        _blocks.add(self)

    def take(self):
        """The error this block holds (and clears), or None.  A host memory read: covers launches that have completed."""
        if not self.np.any():
            return None
        words = self.t.tolist()
        self.t.zero_()
        return _range_error(words)

    def scope(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            status_words()      # the process-wide block exists before any launch, scoped or not
            check(lib().dlip_status_scope(self.t.data_ptr()), "dlip_status_scope")
            try:
                yield self
            finally:
                lib().dlip_status_scope(None)
        return cm()


import weakref as _weakref

_blocks = _weakref.WeakSet()


def check_range(sync: bool = False) -> None:
    """Raise DeepLipRangeError if a kernel reported an activation outside what the split format holds since the last call:
    |v| >= 65520 (infinite in fp16), or a whole produced tensor with its largest magnitude in (0, 2^-2) (lo is subnormal
    there: relative accuracy below fp32 grade; include/deeplip_hip.h).  Without ``sync`` only launches that have completed are
    covered (the call is a host memory read); callers that are about to consume results synchronise first (or pass sync=True).
    Covers the process-wide block and every live StatusBlock (recorded plans report to their own)."""
    t = status_words()
    if sync:
        torch.cuda.synchronize()
    if _status_np.any():
        words = t.tolist()
        t.zero_()
        raise _range_error(words)
    for b in list(_blocks):
        if b.private:
            continue
        err = b.take()
        if err is not None:
            raise err


# ---- low-side range scopes: one evidence word per split-producing launch (dlip_range_scope_*) ----
_SCOPE_SLOTS = 128           # launches per scope (a B = 64 fused step makes ~40)
_EVID_WORDS = 1024           # int32 words of evidence per launch (32 cache lines: DLIP_EVID_WORDS)
_RING_CHUNKS = 16
_rings = {}                  # (device index, stream handle) -> [ring tensor, next chunk]
_scope_depth = threading.local()


def scope_slots(device=None):
    """A zeroed block of evidence words owned by the caller (a StepPlan keeps one for its lifetime)."""
    return torch.zeros(_EVID_WORDS * _SCOPE_SLOTS, dtype=torch.int32, device=device if device is not None else "cuda")


class range_scope:
    """``with range_scope():`` around the launches of one forward pass; the verdict kernel goes out on the current stream at
    exit (join side streams first).  Nested scopes fold into the outermost one of the thread.  Eager scopes draw their words from
    a ring of chunks OF THE STREAM THEY OPEN ON (a scope's verdict kernel re-zeroes its chunk on that stream, and the next user of
    the chunk launches on the same stream: in order, so a chunk is clean before it is written again -- one ring shared by all
    streams had no such order between, say, an ExtractPipeline's run stream and the caller's); a StepPlan passes its own block,
    which the recorded launches then address for the plan's lifetime."""

    def __init__(self, slots=None):
        self.slots = slots
        self.outer = False

    def __enter__(self):
        d = getattr(_scope_depth, "n", 0)
        self.outer = d == 0
        if not self.outer:
            _scope_depth.n = d + 1
            return self
        if _status is None:
            status_words()
        slots = self.slots
        if slots is None:
            key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
            with _lock:
                ring = _rings.get(key)
                if ring is None:
                    # zeroed on the current stream = the stream of every later user of this ring
                    ring = _rings[key] = [torch.zeros(_RING_CHUNKS * _EVID_WORDS * _SCOPE_SLOTS, dtype=torch.int32, device="cuda"), 0]
                slots = ring[0][ring[1] * _EVID_WORDS * _SCOPE_SLOTS:(ring[1] + 1) * _EVID_WORDS * _SCOPE_SLOTS]
                ring[1] = (ring[1] + 1) % _RING_CHUNKS
        check(lib().dlip_range_scope_begin(slots.data_ptr(), min(_SCOPE_SLOTS, slots.numel() // _EVID_WORDS)), "dlip_range_scope_begin")
        _scope_depth.n = d + 1          # only once the scope is really open: a failed begin leaves the thread's depth as it was
        return self

    def __exit__(self, et, ev, tb):
        _scope_depth.n -= 1
        if self.outer:
            rc = lib().dlip_range_scope_end(torch.cuda.current_stream().cuda_stream)
            if et is None:
                check(rc, "dlip_range_scope_end")
        return False


def scoped_eval(fn):
    """Method decorator: an eval-mode call runs inside a low-side range scope (train mode is left alone: its split operands
    carry explicit power-of-two scales, deeplip_amd/autograd*.py)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        if getattr(self, "training", False):
            return fn(self, *a, **k)
        with range_scope():
            return fn(self, *a, **k)
    return wrapper
