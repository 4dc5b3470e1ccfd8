"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950,
loads without a GPU and exports every symbol include/probpose_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "probpose_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built_lib):
    from probpose_pytorch_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 10
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/probpose_hip.h but not exported"
    # the Python binding covers the whole header too
    assert sorted(_lib.EXPORTS) == declared


def test_version_and_error_string(built_lib):
    assert built_lib.pp_version() >= 100
    assert isinstance(built_lib.pp_last_error(), bytes)


def test_argument_validation_without_gpu(built_lib):
    """Host-side checks run before any launch, so they can be exercised on CPU."""
    from probpose_pytorch_amd import _lib
    rc = built_lib.pp_decode_f32(None, None, None, None, None, 1, 17, 64, 48, None, None,
                                 1.0, 1.0, 1.0, 1.0, None, None, None, None, None, None, None, None, 0, None)
    assert rc != 0 and b"null" in built_lib.pp_last_error()
    assert built_lib.pp_decode_workspace_bytes(1, 17, 64, 48) == (17 + 2) * 4     # hand-over list of the wave-per-map path
    assert built_lib.pp_decode_workspace_bytes(2, 133, 96, 72) == (2 * 133 + 2) * 4
    assert built_lib.pp_decode_workspace_bytes(1, 17, 32, 24) == 0                # fits LDS, no list
    assert built_lib.pp_decode_workspace_bytes(1, 20, 256, 256) == 20 * 256 * 256 * 12
    with pytest.raises(_lib.HipExtensionError):
        _lib.check(rc, "pp_decode_f32")


def test_product_path_fails_loudly_without_gpu():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from probpose_pytorch_amd import Codec, ProbMap, _lib, get_heatmap_expected_value
    hm = np.zeros((17, 64, 48), np.float32)
    with pytest.raises(_lib.HipExtensionError):
        get_heatmap_expected_value(hm, np.full(17, 0.05))
    with pytest.raises(_lib.HipExtensionError):
        Codec(ProbMap((192, 256), (48, 64), np.full(17, 0.05))).decode_heatmap(hm)


def test_gemm_argument_validation_without_gpu(built_lib):
    """pp_gemm's host-side refusals come before any launch (no GPU needed): retired tile selectors, the fused final
    layer on a C pointer that cannot take the LDS epilogue (it would otherwise fall through to the direct store loop
    and write 256-channel rows into the [B,K,HW] heat buffer), persistent tile on a residual layer, the four-wave forms
    on what they do not serve, decode flags."""
    import ctypes as C
    from probpose_pytorch_amd import _lib
    a = _lib.GemmArgs()
    a.A, a.W, a.C = 0x1000, 0x2000, 0x3000
    a.M, a.N, a.Kd, a.lda, a.ldw, a.ldc, a.batch, a.dtype = 192, 256, 256, 256, 256, 256, 1, _lib.PP_BF16
    for tile in (11, 12, 21, -1):
        a.tile = tile
        assert built_lib.pp_gemm(C.byref(a), None) != 0 and b"tile" in built_lib.pp_last_error()
    # the four-wave forms: 15 - 17 (per launch) exist in lab builds only; the stream forms 18 - 20 take whole tiles, K >= 512, bf16
    a.tile = 16
    assert built_lib.pp_gemm(C.byref(a), None) != 0 and b"lab builds" in built_lib.pp_last_error()
    a.M, a.N, a.Kd, a.lda, a.ldw, a.ldc = 300, 576, 512, 512, 512, 576
    for tile in (18, 19, 20):
        a.tile = tile
        assert built_lib.pp_gemm(C.byref(a), None) != 0 and b"tiles 18 - 20" in built_lib.pp_last_error()
    a.M, a.Kd, a.lda, a.ldw, a.tile = 768, 256, 256, 256, 19
    assert built_lib.pp_gemm(C.byref(a), None) != 0 and b"K >= 512" in built_lib.pp_last_error()
    a.Kd, a.lda, a.ldw, a.epilogue, a.residual = 512, 512, 512, _lib.EPI_RESIDUAL | _lib.EPI_OUT_F32, 0x3000
    assert built_lib.pp_gemm(C.byref(a), None) != 0 and b"tiles 15 - 20" in built_lib.pp_last_error()
    a.M, a.N, a.Kd, a.lda, a.ldw, a.ldc, a.epilogue, a.residual = 192, 256, 256, 256, 256, 256, 0, None
    a.tile, a.epilogue = 9, _lib.EPI_FUSE_FINAL | _lib.EPI_RELU
    a.final_w, a.final_b, a.hm_K, a.hm_HW, a.hm_temperature = 0x4000, 0x5000, 17, 3072, 0.5
    a.C = 0x3004                                           # 4-byte aligned only: no LDS epilogue possible
    assert built_lib.pp_gemm(C.byref(a), None) != 0 and b"FUSE_FINAL" in built_lib.pp_last_error()
    a.C, a.epilogue, a.tile = 0x3000, _lib.EPI_RESIDUAL | _lib.EPI_OUT_F32, 13
    a.residual = 0x3000
    assert built_lib.pp_gemm(C.byref(a), None) != 0 and b"tile 13" in built_lib.pp_last_error()
    rc = built_lib.pp_decode_f32(0x1000, None, None, None, None, 1, 17, 64, 48, 0x2000, 0x3000,
                                 1.0, 1.0, 1.0, 1.0, None, None, None, None, None, None, None, None, 8, None)
    assert rc != 0 and b"flags" in built_lib.pp_last_error()
