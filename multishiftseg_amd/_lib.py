"""ctypes binding of libmss_hip.so (include/mss_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call returns
non-zero, this module raises (the reference silently fell back to a slow PyTorch path inside a
bare ``except``: ops/modules/ms_deform_attn.py:116-121).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_double, c_float, c_int, c_longlong, c_uint32, c_void_p

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so libmss_hip.so binds to the same runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSS_LIB", os.path.join(_HERE, "libmss_hip.so"))   # MSS_LIB: A/B experiments only

MSS_ABI_VERSION = 8          # include/mss_hip.h
MSS_ERR_BAD_ARG = 1001
MSS_ERR_UNSUPPORTED = 1002


class MssError(RuntimeError):
    pass


MSS_OODM_BATCH = 16            # include/mss_hip.h


class MssOodmBatch(Structure):
    """include/mss_hip.h: up to 16 (score, label, keys, lane_counts, n) entries of mss_oodm_compact_lanes_batch_f32."""
    _fields_ = [("score", c_void_p * MSS_OODM_BATCH), ("label", c_void_p * MSS_OODM_BATCH), ("keys", c_void_p * MSS_OODM_BATCH),
                ("lane_counts", c_void_p * MSS_OODM_BATCH), ("n", ctypes.c_longlong * MSS_OODM_BATCH)]


class MssConvArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("w", c_void_p), ("y", c_void_p),
        ("in_scale", c_void_p), ("in_shift", c_void_p),
        ("out_scale", c_void_p), ("out_shift", c_void_p),
        ("res", c_void_p),
        ("N", c_int), ("H", c_int), ("W", c_int), ("C", c_int), ("ldx", c_int),
        ("OH", c_int), ("OW", c_int), ("K", c_int), ("Kpad", c_int), ("ldy", c_int),
        ("R", c_int), ("S", c_int), ("stride", c_int), ("dil", c_int), ("pad", c_int),
        ("in_ss_stride", c_int), ("in_relu", c_int), ("out_relu", c_int),
        ("ldres", c_int),
        ("M", c_int), ("mtiles", c_int), ("ntiles", c_int),
        ("batch", c_int), ("x_bs", c_longlong), ("w_bs", c_longlong), ("y_bs", c_longlong),
        ("stats", c_void_p),
        ("res_mask", c_int),
        ("w_split", c_void_p),
        ("route", c_int),
    ]


class MssRclArgs(Structure):
    _fields_ = [
        ("logit", c_void_p), ("score", c_void_p), ("target", c_void_p),
        ("B", c_int), ("C", c_int), ("H", c_int), ("W", c_int),
        ("w_ce_orig", c_float), ("w_ce_aug", c_float), ("w_contras", c_float),
        ("m0", c_float), ("m1", c_float), ("m2", c_float),
        ("select", c_int), ("selection_ratio", c_float),
    ]


P = c_void_p
I = c_int
L = c_longlong
F = c_float
U = c_uint32

# name -> argtypes, exactly the declarations of include/mss_hip.h
SIGNATURES = {
    "mss_abi_version": [],
    "mss_env_reset": [],
    "mss_env_generation": [],
    "mss_msda_forward_f32": [P, P, P, P, P, I, I, I, I, I, I, I, P, P],
    "mss_msda_forward_f64": [P, P, P, P, P, I, I, I, I, I, I, I, P, P],
    "mss_msda_backward_f32": [P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, P, P],
    "mss_msda_backward_f64": [P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, P, P],
    "mss_msda_backward_workspace_bytes": [P, I, I, I, I, I, I],
    "mss_msda_backward_binned_f32": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, P, P, L, P],
    "mss_msda_backward_binned_proj_f32": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, L, P, L, P, L, P],
    "mss_msda_forward_fused_f32": [P, P, P, P, P, P, I, I, I, I, I, I, I, P, P],
    "mss_msda_prepare_f32": [P, P, P, P, I, I, I, I, I, P, P, P],
    "mss_msda_prepare_backward_f32": [P, P, P, P, I, I, I, I, I, P, P, P],
    "mss_msda_prepare_backward_ld_f32": [P, P, P, P, I, I, I, I, I, P, L, P, L, P],
    "mss_msda_forward_fused_ld_f32": [P, P, P, P, L, P, L, P, I, I, I, I, I, I, I, P, P],
    "mss_msda_forward_fused_save_f32": [P, P, P, P, L, P, L, P, I, I, I, I, I, I, I, P, P, P, P],
    "mss_msda_prepare_ld_f32": [P, L, P, L, P, P, I, I, I, I, I, P, P, P],
    "mss_conv2d_forward_f32": [POINTER(MssConvArgs), P],
    "mss_conv2d_kpad": [I],
    "mss_conv2d_forward_route": [POINTER(MssConvArgs)],
    "mss_gemm_split_last_mfma": [],
    "mss_gemm_split_weights_bytes": [I, I, I],
    "mss_gemm_split_weights_bf16x3": [P, P, I, I, I, L, P],
    "mss_conv_split_weights_bf16x3": [P, P, I, I, I, P],
    "mss_conv2d_pack_weights_f32": [P, P, I, I, I, I, I, I, I, P],
    "mss_conv2d_wgrad_workspace_bytes": [POINTER(MssConvArgs), I],
    "mss_conv2d_wgrad_f32": [POINTER(MssConvArgs), P, I, P, I, P, L, P],
    "mss_conv2d_wgrad_route": [POINTER(MssConvArgs), I],
    "mss_conv2d_unpack_wgrad_f32": [P, P, I, I, I, I, I, I, I, P],
    "mss_nchw_to_nhwc_pad_f32": [P, P, I, I, I, I, I, P],
    "mss_im2col3x3_c3_f32": [P, P, I, I, I, P],
    "mss_stem_conv_pool_f32": [P, P, P, I, I, I, I, P],
    "mss_bn_stats_nhwc_f32": [P, L, I, I, P, P],
    "mss_bn_finalize_train_f32": [P, L, I, P, P, F, F, P, P, P, P, P, P, P],
    "mss_bn_fold_train_from_partials_f32": [P, L, I, P, L, P, P, F, F, P, P, P, P, P, P, P],
    "mss_bn_fold_eval_f32": [P, P, P, P, F, I, P, P, P],
    "mss_affine_relu_nhwc_f32": [P, I, P, I, L, I, P, P, I, P],
    "mss_bn_relu_bwd_reduce_f32": [P, I, P, I, L, I, P, P, P, P, I, P, P],
    "mss_bn_relu_bwd_apply_f32": [P, I, P, I, P, I, L, I, P, P, P, P, P, I, P, P, P, P],
    "mss_maxpool3s2_nhwc_f32": [P, I, P, I, I, I, I, I, I, I, P],
    "mss_col_reduce_accum_doubles": [L, I],
    "mss_colsum_workspace_floats": [I, I, I],
    "mss_gap_nhwc_f32": [P, I, P, I, I, I, P, P],
    "mss_gap_from_partials_f32": [P, I, I, I, P, P],
    "mss_broadcast_rows_nhwc_f32": [P, P, I, I, I, I, P, P, I, P],
    "mss_colsum_nhwc_f32": [P, I, P, I, I, I, P, P],
    "mss_upsample_ac_nhwc_f32": [P, I, P, I, I, I, I, I, I, I, P],
    "mss_upsample_ac_nhwc_bwd_f32": [P, I, P, I, I, I, I, I, I, I, P],
    "mss_ood_score_f32": [P, I, P, I, I, I, I, I, I, I, P, P, P, P],
    "mss_ood_score_bwd_f32": [P, I, P, P, I, I, I, I, I, I, P, I, P, I, P],
    "mss_m2f_score_f32": [P, P, I, I, I, I, I, I, I, P, P],
    "mss_rcl_pass1_f32": [POINTER(MssRclArgs), P, P, P, P, P, P],
    "mss_rcl_select_f32": [P, L, P, F, P, P, P],
    "mss_rcl_select_merged_f32": [P, L, P, F, P, I, P, P],
    "mss_rcl_pass2_f32": [POINTER(MssRclArgs), P, P, P, P, P, F, P, P],
    "mss_rcl_num_compact_blocks": [I, I, I],
    "mss_rcl_compact_f32": [P, I, I, I, P, P, P, P, P, P],
    "mss_rcl_cin_bwd_f32": [POINTER(MssRclArgs), P, P, F, P, P],
    "mss_rcl_pairs_f32": [P, P, P, P, P, L, F, P, I, F, P, P],
    "mss_rcl_pairs_device_f32": [P, P, P, P, I, L, U, U, F, P, I, F, P, P],
    "mss_rcl_pairs_device2_f32": [P, P, P, P, P, L, U, U, U, F, F, P, F, P, P],
    "mss_rcl_workspace_bytes": [I, I, I],
    "mss_rcl_loss_device_f32": [POINTER(MssRclArgs), P, L, L, U, P, P, P, P],
    "mss_rcl_finalize_f32": [POINTER(MssRclArgs), P, P, P, P],
    "mss_rcl_select_init_f32": [P, F, P, P, P],
    "mss_rcl_select_hist_f32": [P, L, P, I, P, P],
    "mss_rcl_select_pick_f32": [P, P, I, P],
    "mss_rcl_pairs_global_f32": [P, P, U, U, U, P, P, I, U, U, U, U, F, P, I, F, P, P, P],
    "mss_rcl_gather_f32": [P, P, U, P, P],
    "mss_rcl_scatter_add_f32": [P, P, U, P, P],
    "mss_adam_step_f32": [P, P, P, P, L, c_double, c_double, c_double, c_double, c_double, I, P],
    "mss_wino_num_tiles": [I, I, I, I, I],
    "mss_wino_pack_weights_f32": [P, P, I, I, I, I, I, P],
    "mss_wino_pack_split_bf16x3": [P, P, I, I, I, I, P],
    "mss_wino_input_transform_f32": [P, I, I, I, I, I, I, I, P, P, I, P, P],
    "mss_wino_input_transform_bnbwd_f32": [P, I, P, I, I, I, I, I, I, I, P, P, P, P, P, I, P, P],
    "mss_wino_input_transform_upcat_f32": [P, I, I, P, I, I, I, I, I, I, I, I, P, P],
    "mss_wino_input_transform_aspp3_f32": [P, I, I, I, I, I, I, P, P, P, P, P],
    "mss_wino_output_transform_f32": [P, I, I, I, I, I, I, P, I, P, I, P, P],
    "mss_wino_output_stats_parts": [I, I, I, I, I, I],
    "mss_bn_stats_partials_f32": [P, L, I, P, P],
    "mss_wino_grad_output_transform_f32": [P, I, I, I, I, I, I, I, P, P],
    "mss_wino_weight_grad_transform_f32": [P, P, I, I, I, I, I, P],
    "mss_m2f_fused_score_f32": [P, P, I, I, I, I, I, I, I, I, I, I, P, P],
    "mss_m2f_fused_score_ws_f32": [P, P, I, I, I, I, I, I, I, I, I, I, P, P, P],
    "mss_oodm_compact_f32": [P, P, L, L, L, P, P, P],
    "mss_oodm_compact_packed_f32": [P, P, L, L, L, P, P, P],
    "mss_oodm_compact_lanes_f32": [P, P, L, L, L, P, P, P],
    "mss_oodm_compact_lanes_batch_f32": [P, I, L, L, P],
    "mss_oodm_gather_lanes_u32": [P, L, P, P, P, P],
    "mss_oodm_sort_temp_bytes": [L],
    "mss_oodm_compact_lanes_cap": [L],
    "mss_oodm_sort_u32": [P, P, L, P, L, P],
    "mss_oodm_rank_blocks": [L],
    "mss_oodm_measures_f64": [P, L, P, L, c_double, P, P, P, P],
    "mss_add_layernorm_f32": [P, P, L, I, P, P, F, P, P, P],
    "mss_add_layernorm_bwd_workspace_floats": [L, I],
    "mss_add_layernorm_bwd_f32": [P, P, P, P, L, I, P, P, P, P, P, P],
    "mss_add_layernorm_bwd_sum_f32": [P, P, P, P, L, I, P, P, P, P, P, P, P],
    "mss_add_layernorm_q_f32": [P, P, L, I, P, P, F, P, P, P, L, P, P],
    "mss_add_layernorm_bwd_sum2_f32": [P, P, P, P, P, L, I, P, P, P, P, P, P, P],
    "mss_groupnorm_workspace_floats": [I, I, I, I],
    "mss_groupnorm_nhwc_f32": [P, I, L, I, I, I, I, P, P, F, I, P, I, L, P, P],
    "mss_groupnorm_stat_offset": [I, I, I],
    "mss_groupnorm_bwd_workspace_floats": [I, I, I, I],
    "mss_groupnorm_nhwc_bwd_f32": [P, I, L, P, I, L, I, I, I, I, P, P, P, I, P, I, P, P, P, P],
    "mss_upsample_bilinear_bwd_nhwc_f32": [P, I, I, I, I, P, I, L, I, I, I, I, P],
    "mss_nchw_to_nhwc_strided_f32": [P, I, I, I, P, I, L, I, P],
    "mss_upsample_bilinear_add_nhwc_f32": [P, I, L, I, I, I, P, I, P, I, I, I, I, P],
    "mss_nhwc_to_nchw_f32": [P, I, L, I, I, I, P, P],
    "mss_data_pair_f32": [P, P, P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, I, I, P, P, P],
    "mss_peak_mfma_f32": [P, I, I, P],
    "mss_peak_mfma_bf16": [P, I, I, I, P],
    "mss_peak_clock": [P, ctypes.c_ulonglong, P],
    "mss_peak_stream_f32": [P, P, L, I, P],
    "mss_peak_scatter_f32": [P, P, L, I, I, L, P],
}
# entry points that return a plain value rather than a status code
_VALUE_RETURNING = {"mss_gemm_split_last_mfma", "mss_conv2d_wgrad_route", "mss_gemm_split_weights_bytes", "mss_abi_version", "mss_env_reset", "mss_env_generation", "mss_rcl_workspace_bytes", "mss_msda_backward_workspace_bytes", "mss_conv2d_kpad", "mss_conv2d_forward_route", "mss_rcl_num_compact_blocks", "mss_wino_num_tiles",
                    "mss_oodm_sort_temp_bytes", "mss_oodm_compact_lanes_cap", "mss_oodm_rank_blocks", "mss_wino_output_stats_parts",
                    "mss_conv2d_wgrad_workspace_bytes", "mss_col_reduce_accum_doubles", "mss_colsum_workspace_floats",
                    "mss_add_layernorm_bwd_workspace_floats", "mss_groupnorm_workspace_floats",
                    "mss_groupnorm_stat_offset", "mss_groupnorm_bwd_workspace_floats"}
_RETURNS_LONGLONG = {"mss_oodm_compact_lanes_cap", "mss_gemm_split_weights_bytes", "mss_rcl_workspace_bytes", "mss_msda_backward_workspace_bytes", "mss_wino_num_tiles", "mss_oodm_sort_temp_bytes", "mss_conv2d_wgrad_workspace_bytes",
                     "mss_col_reduce_accum_doubles", "mss_colsum_workspace_floats",
                     "mss_add_layernorm_bwd_workspace_floats", "mss_groupnorm_workspace_floats",
                     "mss_groupnorm_stat_offset", "mss_groupnorm_bwd_workspace_floats"}

_lib = None


def load():
    """dlopen libmss_hip.so (once). Raises MssError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MssError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C multishiftseg_amd/csrc` (there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = c_longlong if name in _RETURNS_LONGLONG else c_int
    got = lib.mss_abi_version()
    if got != MSS_ABI_VERSION:
        raise MssError(f"{LIB_PATH} has ABI version {got}, this package expects {MSS_ABI_VERSION}: rebuild it "
                       "(`make -C multishiftseg_amd/csrc` or __graft_entry__.build())")
    _lib = lib
    return lib


def reset_env_cache():
    """The library caches the MSS_* environment switches per call site; call this after changing one mid-process (tests,
    A/B tools). A no-op when the library has not been loaded yet (its first read will see the current environment)."""
    if _lib is not None:
        _lib.mss_env_reset()


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The current torch stream of the current device as a raw hipStream_t. Through torch._C this costs ~0.3 us; building a
    torch.cuda.Stream object per launch (`torch.cuda.current_stream().cuda_stream`) cost ~3 us of the ~10 us a launch takes
    from Python -- and the one-image eval forward and the metric updates are bound by exactly that."""
    if _raw_stream is not None:
        return c_void_p(_raw_stream(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


_fns = {}


def call(name, *args):
    """Call a status-returning entry point on the current torch stream; raise on non-zero."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args, stream_ptr())
    if rc != 0:
        kind = {MSS_ERR_BAD_ARG: "bad argument", MSS_ERR_UNSUPPORTED: "unsupported shape"}.get(rc, "hipError_t")
        raise MssError(f"{name} failed with code {rc} ({kind})")


def status(name, *args):
    """As call(), but hands the status code back instead of raising (callers that have a fallback for MSS_ERR_UNSUPPORTED)."""
    return getattr(load(), name)(*args, stream_ptr())


def value(name, *args):
    assert name in _VALUE_RETURNING
    return getattr(load(), name)(*args)
