"""ctypes binding of lib/libprobpose_hip.so (C ABI: include/probpose_hip.h)."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libprobpose_hip.so")

PP_F32, PP_BF16, PP_FP8 = 0, 1, 2
PP_MAX_RADIUS = 9
PP_MAX_TAPS = 2 * PP_MAX_RADIUS + 1
EPI_BIAS, EPI_GELU, EPI_RELU, EPI_RESIDUAL, EPI_OUT_F32, EPI_ROWBIAS, EPI_HEATMAP = 1, 2, 4, 8, 16, 32, 64
DECODE_NO_WAVE, DECODE_SCREEN, DECODE_ALL_PIXEL, DECODE_WAVE = 1, 2, 4, 16
EPI_HEADMAJOR = 4096
EPI_OUT_FP8, EPI_NOCLAMP, EPI_FUSE_FINAL = 512, 1024, 2048      # 128 / 256: retired (LayerNorm fusion, round 1)

_lock = threading.Lock()
_lib = None


class HipExtensionError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("residual", C.c_void_p), ("rowbias", C.c_void_p),
        ("rowoff", C.c_void_p), ("out_rowmap", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("Kd", C.c_int),
        ("lda", C.c_int), ("ldw", C.c_int), ("ldc", C.c_int),
        ("seg_len", C.c_int), ("rowbias_period", C.c_int), ("batch", C.c_int),
        ("strideA", C.c_longlong), ("strideW", C.c_longlong), ("strideC", C.c_longlong),
        ("strideBias", C.c_longlong), ("strideRowoff", C.c_longlong), ("strideRowmap", C.c_longlong),
        ("dtype", C.c_int), ("epilogue", C.c_int),
        ("hm_K", C.c_int), ("hm_HW", C.c_int), ("hm_temperature", C.c_float),
        ("C2", C.c_void_p), ("ldc2", C.c_int), ("stats_out", C.c_void_p), ("stats_in", C.c_void_p),
        ("stats_parts", C.c_int), ("colsum", C.c_void_p), ("ln_eps", C.c_float), ("tile", C.c_int),
        ("out_scale", C.c_float),
        ("splitk", C.c_int),
        ("strideA_k", C.c_longlong), ("strideW_k", C.c_longlong), ("strideC_k", C.c_longlong),
        ("strideRowoff_k", C.c_longlong),
        ("final_w", C.c_void_p), ("final_b", C.c_void_p),
    ]


_vp, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
_SIGNATURES = {
    "pp_version": (C.c_int, []),
    "pp_last_error": (C.c_char_p, []),
    "pp_device_ok": (C.c_int, []),
    "pp_decode_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "pp_decode_f32": (C.c_int, [_vp] * 5 + [_i] * 4 + [_vp, _vp] + [_d] * 4 + [_vp] * 8 + [_i, _vp]),
    "pp_gemm": (C.c_int, [C.POINTER(GemmArgs), _vp]),
    "pp_layernorm": (C.c_int, [_vp, _vp, _vp, _f, _i, _i, _vp, _i, _vp]),
    "pp_layernorm_fp8": (C.c_int, [_vp, _vp, _vp, _f, _i, _i, _vp, _f, _vp]),
    "pp_attention": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pp_attention_headmajor": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pp_attention_fp8out": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "pp_patchify": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pp_maxpool_relu": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pp_maxpool_relu_sum": (C.c_int, [_vp, _i, C.c_longlong, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pp_final_heatmap": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "pp_final_logits": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "pp_sparsemax_rows": (C.c_int, [_vp, C.c_longlong, _i, _f, _vp]),
    "pp_dark_decode_lds_bytes": (C.c_size_t, [_i, _i, _i]),
    "pp_dark_decode_f32": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _i, _d, _d, _vp, _vp, _vp, _vp]),
    "pp_aux_tail": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pp_tokens_to_nchw": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pp_nchw_to_tokens": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pp_encode_probmaps": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pp_heatmap_argmax": (C.c_int, [_vp, C.c_longlong, _i, _i, _vp, _vp, _vp]),
    "pp_pck_counts": (C.c_int, [_vp, _vp, _i, _vp, _vp, _vp, _d, _i, _i, _vp, _vp, _vp]),
    "pp_frontend_plan_bytes": (C.c_longlong, [_i, _vp, _i, _i]),
    "pp_frontend_plan_build": (C.c_int, [_i, _vp, _i, _i, _vp, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "pp_frontend_crop_resize": (C.c_int, [_vp, _i, _i, C.c_longlong, _vp, _i, _i, C.c_longlong, _i, _i, _vp, _vp]),
}
EXPORTS = tuple(_SIGNATURES)


def lib():
    """Load the extension once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise HipExtensionError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
            h = C.CDLL(LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(h, name)  # AttributeError => header/library mismatch
                fn.restype, fn.argtypes = res, args
            _lib = h
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().pp_last_error().decode("utf-8", "replace")
        raise HipExtensionError(f"{what or 'libprobpose_hip'} failed ({rc}): {msg}")


def require_device(t=None) -> None:
    """The product path is HIP-only: refuse CPU tensors / missing GPU."""
    import torch
    if not torch.cuda.is_available():
        raise HipExtensionError("no HIP device visible: the ProbPose hot path has no CPU fallback")
    if t is not None and not t.is_cuda:
        raise HipExtensionError("expected a tensor on the GPU (cuda/HIP device)")


def ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
