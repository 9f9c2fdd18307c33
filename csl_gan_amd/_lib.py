"""ctypes binding of libcslgan_hip.so (declared in include/cslgan.h).

The product path has no CPU fallback for CUDA/HIP tensors: if the shared library is missing
or fails to load, :func:`lib` raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSLGAN_LIB_PATH") or os.path.join(_HERE, "libcslgan_hip.so")      # override: kernel experiments
MAX_SEGS = 16
ABI_VERSION = 6          # include/cslgan.h CSLGAN_ABI_VERSION

EXPORTS = [
    "cslgan_version", "cslgan_last_error", "cslgan_last_kernel", "cslgan_device_count",
    "cslgan_sample_sqnorm_f32", "cslgan_sample_sqnorm_bf16", "cslgan_clip_factors_f32", "cslgan_clip_accum_noise_f32",
    "cslgan_clip_accum_noise_bf16", "cslgan_conv2d_wgrad_grouped_bf16out_f32",
    "cslgan_l2_clip_rows_f32", "cslgan_row_l2norm_f32", "cslgan_row_l2norm_bwd_f32", "cslgan_mean_sample_f32",
    "cslgan_conv2d_fwd_f32", "cslgan_conv2d_fwd_x3_f32", "cslgan_conv2d_dgrad_f32", "cslgan_conv2d_wgrad_grouped_f32",
    "cslgan_conv2d_wgrad_scaled_f32", "cslgan_conv2d_wgrad_blocks_f32", "cslgan_conv2d_wgrad_sqnorm_gram_f32", "cslgan_conv2d_wgrad_skinny_f32", "cslgan_conv2d_s2_fwd_f32",
    "cslgan_conv2d_dgrad_x3_f32", "cslgan_conv2d_s2_fwd_x3_f32", "cslgan_u8_to_f32_nhwc", "cslgan_split_filter_x3_f32",
    "cslgan_depth_to_space_f32", "cslgan_fold_channels4_f32",
    "cslgan_bias_grad_grouped_f32", "cslgan_act_bwd_f32", "cslgan_groupnorm_act_f32", "cslgan_groupnorm_apply_parts_f32", "cslgan_groupnorm_affine_parts_f32", "cslgan_batchnorm_act_f32", "cslgan_batchnorm_eval_act_f32",
    "cslgan_norm_act_bwd_f32", "cslgan_norm_bwd_ws_floats",
    "cslgan_adam_step_f32", "cslgan_adam_step_dev_f32", "cslgan_adam_multi_f32",
    "cslgan_adaptive_clip_f32", "cslgan_segment_means_f32", "cslgan_segment_means_bwd_f32", "cslgan_dstep_stats_f32", "cslgan_grad_log_stats_f32",
    "cslgan_lerp_rows_f32", "cslgan_lipschitz_term_f32", "cslgan_lipschitz_term_bwd_f32",
    "cslgan_conv2d_fwd_bf16s", "cslgan_conv2d_dgrad_bf16s", "cslgan_conv2d_wgrad_grouped_bf16s", "cslgan_cast_f32_bf16", "cslgan_cast_bf16_f32",
    "cslgan_act_bwd_bf16", "cslgan_bias_grad_grouped_bf16", "cslgan_linear_k1_dgrad_bf16s", "cslgan_linear_k1_wgrad_bf16s",
    "cslgan_conv2d_c3_fwd_bf16out", "cslgan_conv2d_c3_wgrad_bf16gy", "cslgan_groupnorm_act_bf16s",
    "cslgan_conv2d_fwd_skinny_bf16in", "cslgan_conv2d_dgrad_skinny_bf16in", "cslgan_conv2d_wgrad_scaled_bf16s",
]


class SegsT(C.Structure):
    _fields_ = [
        ("n_seg", C.c_int32), ("_pad", C.c_int32),
        ("inp", C.c_void_p * MAX_SEGS), ("out", C.c_void_p * MAX_SEGS), ("noise", C.c_void_p * MAX_SEGS),
        ("len", C.c_int64 * MAX_SEGS), ("row_stride", C.c_int64 * MAX_SEGS), ("rows", C.c_int64 * MAX_SEGS), ("call_counter", C.c_void_p),
    ]


MAX_CLIP_LAYERS, MAX_CLIP_JOBS = 32, 16   # include/cslgan.h CSLGAN_MAX_CLIP_LAYERS / CSLGAN_MAX_CLIP_JOBS


class AdaptiveClipT(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("n_mat", C.c_int32), ("n_jobs", C.c_int32), ("_pad", C.c_int32),
        ("sq_adapt", C.c_void_p * MAX_CLIP_LAYERS), ("sq_rows", C.c_void_p * MAX_CLIP_LAYERS), ("mat_layer", C.c_int32 * MAX_CLIP_LAYERS),
        ("job_dst", C.c_void_p * MAX_CLIP_JOBS), ("job_first", C.c_int64 * MAX_CLIP_JOBS), ("job_layer", C.c_int32 * MAX_CLIP_JOBS),
        ("job_count", C.c_int32 * MAX_CLIP_JOBS), ("job_scale", C.c_float * MAX_CLIP_JOBS),
    ]


class ConvT(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "compute", "P", "Q")] + [("split_ws", C.c_void_p), ("split_ws_floats", C.c_int64), ("gn_part", C.c_void_p), ("gn_groups", C.c_int32), ("in_relu", C.c_int32), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p)]


_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the C-ABI library; raises HipLibraryMissing if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "libcslgan_hip.so not found at %s — build it with `python -m csl_gan_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for device tensors." % LIB_PATH)
    # torch first: its bundled HIP runtime must be the one already mapped when our library's
    # libamdhip64 dependency is resolved (two HIP runtimes in one process cannot share a device context)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.cslgan_last_error.restype = C.c_char_p
    L.cslgan_last_kernel.restype = C.c_char_p
    vp, i32, i64, f32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
    sig = {
        "cslgan_version": [],
        "cslgan_device_count": [],
        "cslgan_sample_sqnorm_f32": [C.POINTER(SegsT), i64, vp, vp],
        "cslgan_sample_sqnorm_bf16": [C.POINTER(SegsT), i64, vp, vp],
        "cslgan_clip_accum_noise_bf16": [C.POINTER(SegsT), i64, vp, i32, vp, u64, u64, f32, f32, vp],
        "cslgan_conv2d_wgrad_grouped_bf16out_f32": [C.POINTER(ConvT), vp, vp, i32, f32, vp, vp, vp],
        "cslgan_clip_factors_f32": [vp, i32, i64, vp, i32, f32, i64, vp, vp, vp],
        "cslgan_clip_accum_noise_f32": [C.POINTER(SegsT), i64, vp, i32, vp, u64, u64, f32, f32, vp],
        "cslgan_l2_clip_rows_f32": [vp, vp, i64, i64, f32, vp, vp],
        "cslgan_mean_sample_f32": [vp, i32, i32, i64, vp, vp, i64, f32, f32, u64, u64, vp, vp, vp],
        "cslgan_row_l2norm_f32": [vp, i64, i64, vp, vp],
        "cslgan_row_l2norm_bwd_f32": [vp, vp, vp, i64, i64, vp, vp],
        "cslgan_conv2d_fwd_f32": [C.POINTER(ConvT), vp, vp, vp, vp, i32, vp, vp],
        "cslgan_conv2d_fwd_x3_f32": [C.POINTER(ConvT), vp, vp, vp, i32, vp, vp, i32, vp, vp],
        "cslgan_conv2d_dgrad_f32": [C.POINTER(ConvT), vp, vp, vp, i32, vp, vp, vp],
        "cslgan_conv2d_dgrad_x3_f32": [C.POINTER(ConvT), vp, vp, vp, vp, i32, vp, vp, vp],
        "cslgan_norm_act_bwd_f32": [vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, i32, vp, vp, vp, vp, vp],
        "cslgan_depth_to_space_f32": [vp, i32, i32, i32, i32, i32, vp, vp],
        "cslgan_fold_channels4_f32": [vp, i64, i32, i32, vp, vp],
        "cslgan_conv2d_wgrad_grouped_f32": [C.POINTER(ConvT), vp, vp, i32, f32, vp, vp, vp],
        "cslgan_conv2d_wgrad_scaled_f32": [C.POINTER(ConvT), vp, vp, vp, i32, f32, vp, vp],
        "cslgan_conv2d_wgrad_blocks_f32": [C.POINTER(ConvT), vp, vp, f32, i32, C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(vp), vp],
        "cslgan_conv2d_wgrad_sqnorm_gram_f32": [C.POINTER(ConvT), vp, vp, f32, vp, vp],
        "cslgan_conv2d_wgrad_skinny_f32": [C.POINTER(ConvT), vp, vp, f32, vp, i32, vp],
        "cslgan_conv2d_s2_fwd_f32": [C.POINTER(ConvT), vp, vp, vp, i32, vp, i32, vp, vp],
        "cslgan_conv2d_s2_fwd_x3_f32": [C.POINTER(ConvT), vp, vp, vp, vp, i32, vp, i32, vp, vp],
        "cslgan_bias_grad_grouped_f32": [vp, i32, i32, i32, i32, f32, vp, vp, vp],
        "cslgan_act_bwd_f32": [vp, vp, i64, f32, vp, vp],
        "cslgan_groupnorm_act_f32": [vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, i32, vp, vp, vp],
        "cslgan_groupnorm_affine_parts_f32": [vp, i32, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp],
        "cslgan_groupnorm_apply_parts_f32": [vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, i32, vp, vp, i32, vp, vp],
        "cslgan_batchnorm_act_f32": [vp, vp, vp, i64, i32, f32, i32, f32, vp, vp, vp, vp, i64, i32, vp, vp, vp],
        "cslgan_batchnorm_eval_act_f32": [vp, vp, vp, vp, vp, i64, i32, f32, i32, vp, vp, i64, i32, vp, vp],
        "cslgan_adam_step_f32": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp],
        "cslgan_adam_step_dev_f32": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp],
        "cslgan_adam_multi_f32": [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), f32, f32, f32, f32, f32,
                                  i32, vp, vp],
        "cslgan_adaptive_clip_f32": [C.POINTER(AdaptiveClipT), i64, i64, i32, f32, i32, f32, i64, vp, vp, vp, vp, vp, vp],
        "cslgan_segment_means_f32": [vp, i32, C.POINTER(C.c_int32), C.POINTER(f32), vp, vp, vp],
        "cslgan_segment_means_bwd_f32": [vp, vp, i32, C.POINTER(C.c_int32), C.POINTER(f32), vp, vp],
        "cslgan_dstep_stats_f32": [vp, i32, vp, i32, vp, vp, vp, vp, vp],
        "cslgan_grad_log_stats_f32": [vp, i32, i64, i64, i32, vp, i32, f32, vp, vp, vp, vp, vp, vp],
        "cslgan_lerp_rows_f32": [vp, vp, vp, i64, i64, vp, vp],
        "cslgan_lipschitz_term_f32": [vp, i64, i64, i32, f32, vp, vp, vp, vp, vp],
        "cslgan_lipschitz_term_bwd_f32": [vp, vp, vp, vp, i64, i64, i32, f32, vp, vp],
        "cslgan_conv2d_fwd_bf16s": [C.POINTER(ConvT), vp, vp, vp, i32, vp, vp, i32, i32, vp, i32, vp],
        "cslgan_conv2d_dgrad_bf16s": [C.POINTER(ConvT), vp, vp, vp, i32, vp, vp, i32, vp],
        "cslgan_conv2d_wgrad_grouped_bf16s": [C.POINTER(ConvT), vp, vp, i32, f32, vp, i32, vp, vp],
        "cslgan_cast_f32_bf16": [vp, vp, i64, vp],
        "cslgan_u8_to_f32_nhwc": [vp, vp, i32, i32, i32, i32, C.c_float, C.c_float, vp, vp],
        "cslgan_split_filter_x3_f32": [vp, i32, i32, i32, vp, i32, vp],
        "cslgan_cast_bf16_f32": [vp, vp, i64, vp],
        "cslgan_act_bwd_bf16": [vp, vp, i64, f32, vp, vp],
        "cslgan_bias_grad_grouped_bf16": [vp, i32, i32, i32, i32, f32, vp, vp, vp],
        "cslgan_linear_k1_dgrad_bf16s": [vp, vp, vp, i32, i64, vp, vp],
        "cslgan_linear_k1_wgrad_bf16s": [vp, vp, vp, i32, i64, i32, f32, vp, vp, vp],
        "cslgan_conv2d_wgrad_scaled_bf16s": [C.POINTER(ConvT), vp, vp, vp, i32, f32, vp, vp],
        "cslgan_conv2d_c3_fwd_bf16out": [C.POINTER(ConvT), vp, vp, vp, i32, vp, vp],
        "cslgan_conv2d_c3_wgrad_bf16gy": [C.POINTER(ConvT), vp, vp, f32, vp, vp, vp],
        "cslgan_groupnorm_act_bf16s": [vp, i32, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, i32, vp, vp],
        "cslgan_conv2d_fwd_skinny_bf16in": [C.POINTER(ConvT), vp, vp, vp, i32, vp, vp],
        "cslgan_conv2d_dgrad_skinny_bf16in": [C.POINTER(ConvT), vp, vp, vp, i32, vp, vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.cslgan_norm_bwd_ws_floats.argtypes = [i64, i64, i32, i32]
    L.cslgan_norm_bwd_ws_floats.restype = C.c_int64
    if L.cslgan_version() != ABI_VERSION:
        raise HipLibraryMissing("libcslgan_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().cslgan_last_error().decode(errors="replace")
        raise RuntimeError("cslgan %s failed (%d): %s" % (what, rc, msg))
