"""ctypes binding of ``libartalk_hip.so`` (C ABI declared in ``include/artalk_hip.h``).

The library is the product path: if it is missing this module raises, there is no CPU fallback.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C artalk_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

from .config import ARTalkConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARTALK_LIB") or os.path.join(_HERE, "libartalk_hip.so")     # ARTALK_LIB: A/B runs of two builds on one box

OK, EINVAL, EKEY, EMISSING, EHIP, ESTATE, ECAPACITY, EBUSY = 0, -1, -2, -3, -4, -5, -6, -7
DTYPE_F32, DTYPE_I64 = 0, 1

# every symbol include/artalk_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "artalk_create", "artalk_destroy", "artalk_last_error", "artalk_set_tensor", "artalk_finalize_weights",
    "artalk_reserve", "artalk_workspace_bytes", "artalk_weight_bytes", "artalk_infer", "artalk_get_status", "artalk_poll_status", "artalk_last_ticket", "artalk_get_status_of", "artalk_style_encode", "artalk_stream_begin", "artalk_stream_chunk", "artalk_stream_end", "artalk_savgol", "artalk_flame_create", "artalk_flame_verts", "artalk_flame_destroy", "artalk_flame_last_error",
    "artalk_set_profiling", "artalk_get_profile", "artalk_get_kernel_sums", "artalk_set_graphs", "artalk_graph_count", "artalk_set_cu_mask", "artalk_set_audit", "artalk_get_audit", "artalk_calibrate", "artalk_reset_scales", "artalk_get_scales", "artalk_set_tap", "artalk_tap_layout", "artalk_set_precision",
    "artalk_op_gemm", "artalk_op_gemm_ex", "artalk_op_gemm_f16s", "artalk_op_pack_split", "artalk_op_gemm_f16s_packed", "artalk_op_release_scratch", "artalk_op_gemm_p8_plan", "artalk_op_create_masked_stream", "artalk_op_destroy_stream", "artalk_op_mfma_f32_peak", "artalk_op_layernorm", "artalk_op_attention", "artalk_op_w2v_front", "artalk_op_resample_mean", "artalk_op_pool_silu",
    "artalk_op_bsq_history",
]


class ArtalkConfigStruct(C.Structure):
    _fields_ = [
        ("ar_depth", C.c_int32), ("ar_heads", C.c_int32),
        ("vae_depth", C.c_int32), ("vae_heads", C.c_int32), ("vae_hidden", C.c_int32),
        ("code_dim", C.c_int32), ("motion_dim", C.c_int32),
        ("n_levels", C.c_int32), ("patch_nums", C.c_int32 * 8),
        ("w2v_layers", C.c_int32), ("w2v_hidden", C.c_int32), ("w2v_heads", C.c_int32), ("w2v_ffn", C.c_int32),
        ("w2v_n_conv", C.c_int32), ("w2v_conv_kernel", C.c_int32 * 8), ("w2v_conv_stride", C.c_int32 * 8),
        ("w2v_conv_dim", C.c_int32),
        ("w2v_pos_kernel", C.c_int32), ("w2v_pos_groups", C.c_int32),
        ("w2v_ln_eps", C.c_float),
        ("style_dim", C.c_int32), ("style_heads", C.c_int32), ("style_layers", C.c_int32), ("style_ffn", C.c_int32),
        ("style_len", C.c_int32),
    ]


def config_struct(cfg: ARTalkConfig) -> ArtalkConfigStruct:
    w = cfg.w2v
    if w.get("feat_extract_norm", "layer") != "layer" or not w.get("do_stable_layer_norm", True) or not w.get("conv_bias", True):
        raise ValueError("only the XLS-R style wav2vec2 (layer-norm conv stack, stable layer norm, conv bias) is supported")
    if len(set(w["conv_dim"])) != 1:
        raise ValueError("conv_dim must be uniform")
    s = ArtalkConfigStruct()
    s.ar_depth, s.ar_heads = cfg.ar_depth, cfg.ar_heads
    s.vae_depth, s.vae_heads, s.vae_hidden = cfg.vae_depth, cfg.vae_heads, cfg.vae_hidden
    s.code_dim, s.motion_dim = cfg.code_dim, cfg.motion_dim
    s.n_levels = len(cfg.patch_nums)
    for i, p in enumerate(cfg.patch_nums):
        s.patch_nums[i] = p
    s.w2v_layers, s.w2v_hidden = w["num_hidden_layers"], w["hidden_size"]
    s.w2v_heads, s.w2v_ffn = w["num_attention_heads"], w["intermediate_size"]
    s.w2v_n_conv = len(w["conv_kernel"])
    for i, (k, st) in enumerate(zip(w["conv_kernel"], w["conv_stride"])):
        s.w2v_conv_kernel[i], s.w2v_conv_stride[i] = k, st
    s.w2v_conv_dim = w["conv_dim"][0]
    s.w2v_pos_kernel, s.w2v_pos_groups = w["num_conv_pos_embeddings"], w["num_conv_pos_embedding_groups"]
    s.w2v_ln_eps = w["layer_norm_eps"]
    s.style_dim, s.style_heads, s.style_layers = cfg.style_dim, cfg.style_heads, cfg.style_layers
    s.style_ffn, s.style_len = cfg.style_ffn, cfg.style_len
    return s


_lib = None


def lib() -> C.CDLL:
    """Load the shared library once and declare the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.artalk_create.argtypes = [i32, C.POINTER(ArtalkConfigStruct), C.POINTER(vp)]
    L.artalk_create.restype = i32
    L.artalk_destroy.argtypes = [vp]
    L.artalk_destroy.restype = None
    L.artalk_last_error.argtypes = [vp]
    L.artalk_last_error.restype = C.c_char_p
    L.artalk_set_tensor.argtypes = [vp, C.c_char_p, vp, i32, i32, C.POINTER(i64)]
    L.artalk_set_tensor.restype = i32
    L.artalk_finalize_weights.argtypes = [vp]
    L.artalk_finalize_weights.restype = i32
    L.artalk_reserve.argtypes = [vp, i32, i32]
    L.artalk_reserve.restype = i32
    L.artalk_workspace_bytes.argtypes = [vp]
    L.artalk_workspace_bytes.restype = i64
    L.artalk_weight_bytes.argtypes = [vp]
    L.artalk_weight_bytes.restype = i64
    L.artalk_infer.argtypes = [vp, vp, i64, C.POINTER(i64), i32, vp, vp, vp, i64, vp, vp, vp, vp]
    L.artalk_infer.restype = i32
    L.artalk_get_status.argtypes = [vp, C.POINTER(C.c_int), vp]
    L.artalk_get_status.restype = i32
    L.artalk_poll_status.argtypes = [vp, C.POINTER(C.c_int)]
    L.artalk_poll_status.restype = i32
    L.artalk_last_ticket.argtypes = [vp]
    L.artalk_last_ticket.restype = C.c_longlong
    L.artalk_get_status_of.argtypes = [vp, C.c_longlong, C.POINTER(C.c_int)]
    L.artalk_get_status_of.restype = i32
    L.artalk_style_encode.argtypes = [vp, vp, i32, vp, vp]
    L.artalk_style_encode.restype = i32
    L.artalk_stream_begin.argtypes = [vp, i32, vp, vp, vp]
    L.artalk_stream_begin.restype = i32
    L.artalk_stream_end.argtypes = [vp]
    L.artalk_stream_end.restype = i32
    L.artalk_stream_chunk.argtypes = [vp, vp, i64, vp, i64, vp]
    L.artalk_stream_chunk.restype = i32
    L.artalk_flame_create.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, f32, C.POINTER(vp)]
    L.artalk_flame_create.restype = i32
    L.artalk_flame_verts.argtypes = [vp, vp, vp, i32, vp, vp]
    L.artalk_flame_verts.restype = i32
    L.artalk_flame_destroy.argtypes = [vp]
    L.artalk_flame_destroy.restype = None
    L.artalk_flame_last_error.argtypes = [vp]
    L.artalk_flame_last_error.restype = C.c_char_p
    L.artalk_savgol.argtypes = [vp, vp, vp, i32, vp]
    L.artalk_savgol.restype = i32
    L.artalk_set_profiling.argtypes = [vp, i32]
    L.artalk_set_profiling.restype = i32
    L.artalk_get_profile.argtypes = [vp, C.POINTER(C.c_double), i32]
    L.artalk_get_profile.restype = i32
    if hasattr(L, "artalk_get_kernel_sums"):      # (round 5)
        L.artalk_get_kernel_sums.argtypes = [vp, C.POINTER(C.c_double), i32]
        L.artalk_get_kernel_sums.restype = i32
    L.artalk_set_graphs.argtypes = [vp, i32]
    L.artalk_set_graphs.restype = i32
    if hasattr(L, "artalk_graph_count"):      # (round 5; an older build loaded through ARTALK_LIB lacks it)
        L.artalk_graph_count.argtypes = [vp, C.POINTER(C.c_longlong)]
        L.artalk_graph_count.restype = i32
    if hasattr(L, "artalk_set_audit"):      # (an older build loaded through ARTALK_LIB for an A/B run may lack it)
        L.artalk_set_audit.argtypes = [vp, i32]
        L.artalk_set_audit.restype = i32
        L.artalk_get_audit.argtypes = [vp, C.c_char_p, i32, C.POINTER(C.c_float), i32]
        L.artalk_get_audit.restype = i32
    if hasattr(L, "artalk_calibrate"):      # (round 5)
        L.artalk_calibrate.argtypes = [vp, f32]
        L.artalk_calibrate.restype = i32
        L.artalk_reset_scales.argtypes = [vp]
        L.artalk_reset_scales.restype = i32
        L.artalk_get_scales.argtypes = [vp, C.POINTER(C.c_int), i32]
        L.artalk_get_scales.restype = i32
    L.artalk_op_gemm.argtypes = [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.artalk_op_gemm.restype = i32
    L.artalk_set_precision.argtypes = [vp, i32]
    L.artalk_set_precision.restype = i32
    L.artalk_op_gemm_f16s.argtypes = [vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.artalk_op_gemm_f16s.restype = i32
    L.artalk_op_pack_split.argtypes = [vp, vp, i64, i32, vp]
    L.artalk_op_pack_split.restype = i32
    L.artalk_op_gemm_f16s_packed.argtypes = [vp, i32, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.artalk_op_gemm_f16s_packed.restype = i32
    if hasattr(L, "artalk_set_tap"):      # (an older build loaded through ARTALK_LIB for a same-box A/B run lacks the round-4 entry points)
        L.artalk_set_tap.argtypes = [vp, vp, i32, i32]
        L.artalk_set_tap.restype = i32
        L.artalk_tap_layout.argtypes = [C.POINTER(C.c_int64), i32]
        L.artalk_tap_layout.restype = i32
        L.artalk_op_release_scratch.argtypes = []
        L.artalk_op_release_scratch.restype = i32
    L.artalk_set_cu_mask.argtypes = [vp, vp, i32]
    L.artalk_set_cu_mask.restype = i32
    L.artalk_op_create_masked_stream.argtypes = [vp, i32, C.POINTER(vp)]
    L.artalk_op_create_masked_stream.restype = i32
    L.artalk_op_destroy_stream.argtypes = [vp]
    L.artalk_op_destroy_stream.restype = i32
    L.artalk_op_gemm_p8_plan.argtypes = [i32, i32, i32, i32]
    L.artalk_op_gemm_p8_plan.restype = i32
    L.artalk_op_gemm_ex.argtypes = [vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.artalk_op_gemm_ex.restype = i32
    L.artalk_op_mfma_f32_peak.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_double), vp]
    L.artalk_op_mfma_f32_peak.restype = i32
    L.artalk_op_layernorm.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp]
    L.artalk_op_layernorm.restype = i32
    L.artalk_op_attention.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp, i32, vp]
    L.artalk_op_attention.restype = i32
    L.artalk_op_w2v_front.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.artalk_op_w2v_front.restype = i32
    L.artalk_op_resample_mean.argtypes = [vp, i32, i32, vp, i32, i32, i32, vp, i32, vp]
    L.artalk_op_resample_mean.restype = i32
    L.artalk_op_pool_silu.argtypes = [vp, i32, i32, i32, vp, vp]
    L.artalk_op_pool_silu.restype = i32
    L.artalk_op_bsq_history.argtypes = [vp, vp, vp, vp, i32, vp]
    L.artalk_op_bsq_history.restype = i32
    _lib = L
    return L


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
