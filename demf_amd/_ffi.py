"""ctypes binding of libdemf_hip.so (the C ABI declared in include/demf_hip.h).

The reference reaches its native operators through pybind extensions of
mmdet3d/mmcv; this is the equivalent thin layer.  cffi is not installed in the
target image, so the binding is ctypes.  There is no fallback: if the library is
missing, ``load()`` raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEMF_LIB_PATH") or os.path.join(_HERE, "lib", "libdemf_hip.so")     # (A/B builds)

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_ptr = ctypes.c_void_p

# name -> argtypes, exactly mirroring include/demf_hip.h
SIGNATURES = {
    "demf_stream_create_cu_masked": [_ptr, _c_int, _c_int, _ptr],
    "demf_spin_us": [_c_int, _ptr],
    "demf_conv_nhwc_f32": [_c_int] * 9 + [_ptr, _ptr, _c_int, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr],
    "demf_conv_stem7_nhwc4_f32": [_c_int] * 4 + [_ptr, _ptr, _c_int, _ptr, _c_int, _ptr, _ptr],
    "demf_maxpool3x3s2_nhwc_f32": [_c_int] * 4 + [_ptr, _ptr, _ptr],
    "demf_nchw3_to_nhwc4_f32": [_c_int] * 3 + [_ptr, _ptr, _ptr],
    "demf_groupnorm_nhwc_f32": [_c_int] * 4 + [_c_float, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_longlong, _ptr],
    "demf_fps_f32": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "demf_fps_ws_f32": [_c_int, _c_int, _c_int, _ptr, _ptr, ctypes.c_longlong, _ptr, _ptr],
    "demf_ball_query_f32": [_c_int, _c_int, _c_int, _c_float, _c_float, _c_int, _ptr, _ptr,
                            _ptr, _ptr],
    "demf_ball_query_grid_ws": [_c_int, _c_int, _ptr, _ptr],
    "demf_ball_query_grid_f32": [_c_int, _c_int, _c_int, _c_float, _c_int] + [_ptr] * 6,
    "demf_group_points_fwd": [_c_int] * 5 + [_ptr] * 4,
    "demf_group_points_bwd": [_c_int] * 5 + [_ptr] * 4,
    "demf_gather_points_fwd": [_c_int] * 4 + [_ptr] * 4,
    "demf_gather_points_bwd": [_c_int] * 4 + [_ptr] * 4,
    "demf_three_nn_f32": [_c_int] * 3 + [_ptr] * 5,
    "demf_three_nn_weights_f32": [_c_int] * 3 + [_ptr] * 6,
    "demf_three_interpolate_fwd": [_c_int] * 4 + [_ptr] * 5,
    "demf_three_interpolate_bwd": [_c_int] * 4 + [_ptr] * 5,
    "demf_group_concat_cl_fwd": [_c_int] * 8 + [_c_float, _c_int] + [_ptr] * 6,
    "demf_group_concat_cl_bwd": [_c_int] * 8 + [_c_float, _c_int] + [_ptr] * 6,
    "demf_gather_rows_cl_fwd": [_c_int] * 4 + [_ptr] * 4,
    "demf_gather_rows_cl_bwd": [_c_int] * 4 + [_ptr] * 4,
    "demf_three_interpolate_cl_fwd": [_c_int] * 6 + [_ptr] * 5,
    "demf_three_interpolate_cl_bwd": [_c_int] * 6 + [_ptr] * 5,
    "demf_three_interpolate_cat_cl_fwd": [_c_int] * 5 + [_ptr] * 6,
    "demf_maxpool_ns_fwd": [_c_int] * 3 + [_ptr] * 4,
    "demf_maxpool_ns_bwd": [_c_int] * 3 + [_ptr] * 4,
    "demf_colsum_f32": [_c_int] * 3 + [_ptr] * 3,
    "demf_nchw_to_tokens": [_c_int] * 5 + [_ptr] * 4,
    "demf_pyramid_to_tokens": [_c_int] * 4 + [_ptr] * 5,
    "demf_pyramid_to_tokens_bf16": [_c_int] * 4 + [_ptr] * 5,
    "demf_vote_targets": [_c_int] * 4 + [_ptr] * 8,
    "demf_box_extent_count": [_c_int] * 4 + [_ptr] * 8,
    "demf_aligned_nms": [_c_int, _c_int, _c_float] + [_ptr] * 6,
    "demf_proposal_targets": [_c_int] * 4 + [_c_float] * 3 + [_ptr] * 18,
    "demf_gt_prep": [_c_int] * 3 + [_ptr] * 10,
    "demf_pad_gt": [_c_int] * 2 + [_ptr] * 8,
    "demf_query_pos_rows": [_c_int] * 2 + [_ptr] * 4,
    "demf_loss_total": [_c_int] + [_ptr] * 4,
    "demf_loss_total_bwd": [_c_int] + [_ptr] * 4,
    "demf_target_weights": [_c_int] + [_ptr] * 5,
    "demf_rows_ln_pos_f32": [_c_int] * 2 + [_ptr] * 4 + [_c_float] + [_ptr] * 4,
    "demf_rows_gemm_f32": [_c_int] * 3 + [_ptr, ctypes.c_longlong, _ptr, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int,
                           _ptr, ctypes.c_longlong, _ptr, _ptr, _c_float, _ptr, ctypes.c_longlong, _ptr],
    "demf_msda_fwd_raw_f32": [_c_int] * 7 + [_ptr, ctypes.c_longlong, _ptr, _ptr, _ptr, ctypes.c_longlong, _c_int, _c_int,
                              _ptr, _ptr, _ptr],
    "demf_msda_fwd_raw_head_f32": [_c_int] * 4 + [_ptr, ctypes.c_longlong, _ptr, _ptr, _ptr, ctypes.c_longlong, _c_int, _c_int,
                                   _ptr, _ptr, _c_int, _c_int, _ptr],
    "demf_invert_index": [_c_int] * 3 + [_ptr] * 4,
    "demf_invert_index_ws": [_c_int] * 3 + [_ptr] * 5,
    "demf_mlp_gemm_bwd_dw_group": [_c_int, _ptr, _ptr],
    "demf_sa_index_chain": [_c_int] * 3 + [_ptr] * 4,
    "demf_split_points": [ctypes.c_longlong, _c_int] + [_ptr] * 4,
    "demf_group_concat_cl_bwd_gather": [_c_int] * 6 + [_ptr] * 5,
    "demf_group_first_fwd": [_c_int] * 5 + [_c_float, _c_int] + [_ptr] * 5 + [_c_int] + [_ptr] * 4 + [_c_float] * 2 + [_ptr] * 6 + [_ptr],
    "demf_group_first_bwd": [_c_int] * 5 + [_c_float, _c_int] + [_ptr] * 10 + [_c_int, _ptr, _c_int] + [_ptr] * 4,
    "demf_adamw_f32": [ctypes.c_longlong] + [_ptr] * 5 + [_c_float] * 7 + [_c_int, _ptr],
    "demf_multi_copy_sumsq": [_c_int, _ptr, _c_int, _ptr, _ptr],
    "demf_sumsq_f32": [ctypes.c_longlong, _ptr, _ptr, _ptr],
    "demf_adamw_state_f32": [_c_int] + [_ptr] * 9 + [_c_float] * 5 + [_ptr],
    "demf_mlp_gemm_fwd": [_c_int] * 4 + [_ptr] * 6,
    "demf_mlp_gemm_fwd_pool": [_c_int] * 4 + [_ptr] * 5 + [_c_int] + [_ptr] * 5,
    "demf_mlp_gemm_fwd_bn": [_c_int] * 4 + [_ptr] * 7 + [_c_float, _c_float] + [_ptr] * 7,
    "demf_mlp_gemm_fwd_pool_bn": [_c_int] * 4 + [_ptr] * 5 + [_c_int] + [_ptr] * 6 + [_c_float, _c_float] + [_ptr] * 7,
    "demf_mlp_gemm_fwd_bn_st": [_c_int] * 4 + [_ptr] * 7 + [_c_float, _c_float] + [_ptr] * 6 + [_c_int, _ptr],
    "demf_mlp_gemm_fwd_pool_bn_st": [_c_int] * 4 + [_ptr] * 5 + [_c_int] + [_ptr] * 4 + [_c_float, _c_float] + [_ptr] * 6 + [_c_int, _ptr],
    "demf_pool_select": [_c_int] * 2 + [_ptr] * 9,
    "demf_pool_select_slot0": [_c_int] * 2 + [_ptr] * 7,
    "demf_mlp_first_stats": [_c_int] * 2 + [_ptr] * 5 + [_c_float, _c_float] + [_ptr] * 7,
    "demf_mlp_gemm_fwd_bn_x4": [_c_int] * 2 + [_ptr] * 8 + [_c_float, _c_float] + [_ptr] * 7,
    "demf_mlp_bwd_fused_x4": [_c_int] * 3 + [_ptr] * 11,
    "demf_mlp_bwd_pool": [_c_int] * 4 + [_ptr] * 17,
    "demf_mlp_bwd_pool_ws": [_c_int, _ptr],
    "demf_bn_finalize": [_c_int, ctypes.c_longlong] + [_ptr] * 3 + [_c_float, _c_float] + [_ptr] * 7,
    "demf_l2norm_rows_fwd": [_c_int] * 2 + [_ptr] * 4,
    "demf_l2norm_rows_bwd": [_c_int] * 2 + [_ptr] * 5,
    "demf_vote_combine_fwd": [_c_int] * 2 + [_ptr] * 7,
    "demf_vote_combine_bwd": [_c_int] * 2 + [_ptr] * 7,
    "demf_bnrelu_maxpool_fwd": [_c_int] * 3 + [_ptr] * 5,
    "demf_bn_bwd_reduce": [_c_int] * 3 + [_ptr] * 9,
    "demf_bn_bwd_vectors": [_c_int, ctypes.c_longlong] + [_ptr] * 8,
    "demf_mlp_gemm_bwd_dx": [_c_int] * 4 + [_ptr] * 3 + [_c_int] + [_ptr] * 5,
    "demf_mlp_gemm_bwd_dx_first": [_c_int] * 3 + [_ptr] * 10,
    "demf_mlp_first_finish": [_c_int, ctypes.c_longlong] + [_ptr] * 7,
    "demf_mlp_gemm_bwd_dx_red": [_c_int] * 4 + [_ptr] * 3 + [_c_int] + [_ptr] * 9,
    "demf_mlp_gemm_bwd_dx_w": [_c_int] * 4 + [_ptr] * 3 + [_c_int] + [_ptr] * 5,
    "demf_mlp_bwd_fused": [_c_int] * 3 + [_ptr] * 3 + [_c_int] + [_ptr] * 15 + [_c_int, _ptr],
    "demf_mlp_bwd_fused_cols": [_c_int] * 5 + [_ptr] * 3 + [_c_int] + [_ptr] * 14,
    "demf_mlp_gemm_bwd_dx_red_v": [_c_int] * 4 + [_ptr] * 3 + [_c_int] + [_ptr] * 13,
    "demf_bn_bwd_reduce_vectors": [_c_int] * 3 + [_ptr] * 11 + [_c_int, _ptr],
    "demf_mlp_gemm_bwd_dw": [_c_int] * 4 + [_ptr] * 3 + [_c_int] + [_ptr] * 6,
    "demf_mlp_gemm_bwd_dw_ld": [_c_int] * 4 + [_ptr] * 3 + [_c_int] + [_ptr] * 5 + [_c_int, _ptr],
    "demf_head_loss_fwd": [_c_int] * 3 + [_ptr] * 14,
    "demf_head_loss_bwd": [_c_int] * 3 + [_ptr] * 17,
    "demf_vote_loss": [_c_int] * 4 + [_c_float] + [_ptr] * 10,
    "demf_vote_loss_fwd": [_c_int] * 4 + [_c_float] + [_ptr] * 8,
    "demf_msda_fwd_f32": [_c_int] * 7 + [_ptr] * 7,
    "demf_msda_bwd_f32": [_c_int] * 7 + [_ptr] * 10,
    "demf_msda_fwd_bf16": [_c_int] * 7 + [_ptr] * 7,
    "demf_msda_bwd_bf16": [_c_int] * 7 + [_ptr] * 10,
    "demf_gemm_f32": [_ptr, _ptr],
    "demf_gemm_group_f32": [_ptr, _c_int, _ptr],
    "demf_set_compute_dtype": [_c_int],
    "demf_set_f16_terms": [_c_int],
    "demf_get_compute_dtype": [],
    "demf_ctx_push": [_ptr],
    "demf_ctx_pop": [],
    "demf_gemm_f32_ctx": [_ptr] * 3,
    "demf_gemm_group_f32_ctx": [_ptr, _ptr, _c_int, _ptr],
    "demf_mlp_gemm_fwd_ctx": [_ptr] + [_c_int] * 4 + [_ptr] * 6,
    "demf_multi_copy": [_c_int, _ptr, _c_int, _ptr],
    "demf_zero_f32": [ctypes.c_longlong, _ptr, _ptr],
    "demf_add_dropout_ln_fwd": [_c_int] * 2 + [_ptr] * 4 + [_c_float] * 2 + [_ptr, _c_int] + [_ptr] * 4,
    "demf_add_dropout_ln_bwd": [_c_int] * 2 + [_ptr] * 5 + [_c_float, _ptr, _c_int, _ptr, _c_int] + [_ptr] * 4,
    "demf_attn_core_fwd": [_c_int] * 4 + [_ptr, _c_float, _c_float, _ptr, _c_int] + [_ptr] * 5,
    "demf_attn_core_bwd": [_c_int] * 4 + [_ptr] * 4 + [_c_float, _c_float, _ptr, _c_int] + [_ptr] * 2,
    "demf_softmax_dropout_fwd": [_c_int] * 2 + [_ptr, _c_float, _ptr, _c_int] + [_ptr] * 3,
    "demf_softmax_dropout_bwd": [_c_int] * 2 + [_ptr, _c_float, _ptr, _c_int] + [_ptr] * 2,
    "demf_msda_prep_fwd": [_c_int] * 5 + [_ptr] * 10,
    "demf_msda_prep_bwd": [_c_int] * 5 + [_ptr] * 14,
    "demf_rng_advance": [_ptr, _ptr],
    "demf_rng_next": [_ptr, _ptr, _ptr],
    "demf_dropout_mask": [ctypes.c_longlong, _c_float, _ptr, _c_int, _ptr, _ptr],
}


class DwJob(ctypes.Structure):
    """include/demf_hip.h: demf_dw_job (field order and types must match)."""
    _fields_ = [("R", _c_int), ("N", _c_int), ("K", _c_int), ("ldx", _c_int), ("G", _ptr), ("dP", _ptr), ("arg", _ptr),
                ("ns", _c_int), ("Y", _ptr), ("vec6", _ptr), ("Xprev", _ptr), ("prev_scale_shift", _ptr), ("dW", _ptr),
                ("lddw", _c_int)]


class GemmDesc(ctypes.Structure):
    """include/demf_hip.h: demf_gemm_desc (field order and types must match)."""
    _ll = ctypes.c_longlong
    _fields_ = [
        ("M", _c_int), ("N", _c_int), ("K", _c_int), ("batch", _c_int), ("zdiv", _c_int),
        ("splitk", _c_int),
        ("A", _ptr), ("sam", _ll), ("sak", _ll), ("sab", _ll), ("sab2", _ll),
        ("A2", _ptr), ("a2_cols", _c_int),
        ("B", _ptr), ("sbn", _ll), ("sbk", _ll), ("sbb", _ll), ("sbb2", _ll),
        ("B2", _ptr), ("b2_rows", _c_int),
        ("C", _ptr), ("scm", _ll), ("scb", _ll), ("scb2", _ll),
        ("C2", _ptr),
        ("bias", _ptr), ("sbias_b", _ll),
        ("rowscale", _ptr), ("srs_m", _ll), ("srs_b", _ll),
        ("alpha", _c_float),
        ("flags", _c_int),
        ("gate", _ptr), ("sgm", _ll), ("sgb", _ll),
        ("gate_scale", _c_float),
        ("drop_p", _c_float),
        ("rng", _ptr),
        ("op_id", _c_int),
        ("asum", _ptr),
    ]

_lib = None


def load():
    """Load libdemf_hip.so; raises if it has not been built (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). demf_amd has no CPU fallback.")
    # torch first: PyTorch-ROCm bundles its own HIP / HSA runtime; if this library were loaded before
    # it, the process would hold two runtimes and the second one finds no device ("no ROCm-capable
    # device is detected" at the first launch - seen with build() and smoke() in one process)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    lib.demf_version.restype = _c_int
    lib.demf_version.argtypes = []
    lib.demf_last_error.restype = ctypes.c_char_p
    lib.demf_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = _c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an entry point; a non-zero return becomes a RuntimeError with the
    library's thread-local message (the upstream pybind wrappers raise likewise)."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.demf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (code {rc}): {msg}")
