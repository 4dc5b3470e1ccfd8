"""Mirror of the hot-path part of the reference's ``probpose/heatmap.py``.

``get_heatmap_expected_value`` keeps the reference signature
(heatmap.py:291-297) but runs the whole OKS-convolution / argmax / sub-pixel /
gather chain in one fused HIP kernel (csrc/pp_decode.hip).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_TAP_CACHE: dict = {}


def oks_tap_table(K: int, H: int, W: int, sigmas) -> tuple[np.ndarray, np.ndarray]:
    """Host-side replacement of ``_prepare_oks_kernels`` (heatmap.py:170-194).

    The reference builds, per keypoint, exp(-(dx^2+dy^2)/(2s)) over a
    (2r+1)^2 window and normalises it; that kernel is exactly the outer
    product of the normalised 1-D Gaussian returned here, so the device runs
    it as a row pass + column pass.  Returns (taps f64 [K, PP_MAX_TAPS],
    radius i32 [K]).  dtype handling follows the reference: a float32 sigma
    rounds (2*sigma)^2 in float32 before the float64 area factor is applied.
    """
    sig = np.asarray(sigmas)
    if sig.ndim != 1 or sig.shape[0] < K:
        raise ValueError(f"sigmas must hold at least K={K} entries, got shape {sig.shape}")
    area = np.sqrt(H / 1.25 * W / 1.25)
    taps = np.zeros((K, _lib.PP_MAX_TAPS), np.float64)
    radius = np.zeros((K,), np.int32)
    for k in range(K):
        two_sigma_sq = (sig[k] * 2) ** 2
        s = float(np.clip(two_sigma_sq * area * 2, 0.55, 3.0))
        r = int(np.ceil(s * 3))
        off = np.arange(-r, r + 1, dtype=np.float64)
        g = np.exp(-(off * off) / (2 * s))
        taps[k, : 2 * r + 1] = g / g.sum()
        radius[k] = r
    return taps, radius


def _device_taps(K, H, W, sigmas, device):
    sig = np.asarray(sigmas)
    key = (K, H, W, sig.dtype.str, sig.tobytes(), str(device))
    hit = _TAP_CACHE.get(key)
    if hit is None:
        taps, radius = oks_tap_table(K, H, W, sig)
        hit = (torch.from_numpy(taps).to(device), torch.from_numpy(radius).to(device))
        if len(_TAP_CACHE) > 64:
            _TAP_CACHE.clear()
        _TAP_CACHE[key] = hit
    return hit


# which implementation pp_decode_f32 runs: 0 = the default per map size; _lib.DECODE_* force another one (A/B tools and
# the equivalence tests; every form returns identical numbers)
DECODE_FLAGS = 0

# zeroed hand-over lists of the wave-per-map path, one per (device, stream): allocated and zeroed once, every call
# leaves their counters zeroed (include/probpose_hip.h), the address stays stable for graph replay
_DECODE_WS: dict = {}


def _decode_workspace(nbytes: int, dev):
    if nbytes <= 0:
        return None
    key = (str(dev), int(torch.cuda.current_stream(dev).cuda_stream))
    t = _DECODE_WS.get(key)
    if t is None or t.numel() < nbytes:
        if len(_DECODE_WS) > 16:
            _DECODE_WS.clear()
        t = torch.zeros((max(nbytes, 1 << 16),), dtype=torch.uint8, device=dev)
        _DECODE_WS[key] = t
    return t


def decode_on_device(heatmaps: torch.Tensor, sigmas, *, den=None, input_size=None,
                     aux=None, want_conv: bool = False) -> dict:
    """Launch the fused decode on device-resident heatmaps (B,K,H,W) f32.

    Returns a dict of device tensors: locs (B,K,2) f32, scores (B,K) f32 and,
    when ``input_size`` is given, kpts (B,K,2) f64; with ``aux`` = (prob, vis,
    oks, err) tensors of B*K elements also aux (3,B,K) f32 and err (B,K) f64;
    with ``want_conv`` the convolved maps (B,K,H,W) f32.
    """
    _lib.require_device(heatmaps)
    L = _lib.lib()
    if heatmaps.dtype != torch.float32:
        heatmaps = heatmaps.float()
    heatmaps = heatmaps.contiguous()
    B, K, H, W = heatmaps.shape
    dev = heatmaps.device
    taps, radius = _device_taps(K, H, W, sigmas, dev)
    out = {
        "locs": torch.empty((B, K, 2), dtype=torch.float32, device=dev),
        "scores": torch.empty((B, K), dtype=torch.float32, device=dev),
    }
    den_x, den_y, in_w, in_h = 1.0, 1.0, 1.0, 1.0
    if input_size is not None:
        out["kpts"] = torch.empty((B, K, 2), dtype=torch.float64, device=dev)
        den_x, den_y = float(den[0]), float(den[1])
        in_w, in_h = float(input_size[0]), float(input_size[1])
    a = [None] * 4
    if aux is not None:
        a = [t.reshape(-1).contiguous().float() for t in aux]
        for t in a:
            if t.numel() != B * K:
                raise ValueError("auxiliary outputs must hold B*K values (1x1 spatial), "
                                 f"got {t.numel()} for B={B}, K={K}")
        out["aux"] = torch.empty((3, B, K), dtype=torch.float32, device=dev)
        out["err"] = torch.empty((B, K), dtype=torch.float64, device=dev)
        if input_size is not None:
            out["packed"] = torch.empty((B, K, 7), dtype=torch.float64, device=dev)
    if want_conv:
        out["conv"] = torch.empty_like(heatmaps)
    ws_bytes = L.pp_decode_workspace_bytes(B, K, H, W)
    if (H, W) in ((64, 48), (96, 72)):
        ws = _decode_workspace(ws_bytes, dev)                # the self-resetting hand-over list: zeroed once, kept
    else:
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if ws_bytes else None
    from . import ops as _ops
    with torch.cuda.device(dev):
        rc = _ops._timed("decode", float(B * K * (H * W * 4 + 16 + 28)), lambda: L.pp_decode_f32(
            _lib.ptr(heatmaps), _lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(a[3]),
            B, K, H, W, _lib.ptr(taps), _lib.ptr(radius), den_x, den_y, in_w, in_h,
            _lib.ptr(out.get("kpts")), _lib.ptr(out["scores"]), _lib.ptr(out["locs"]),
            _lib.ptr(out.get("aux")), _lib.ptr(out.get("err")), _lib.ptr(out.get("conv")),
            _lib.ptr(out.get("packed")), _lib.ptr(ws), int(DECODE_FLAGS), _lib.stream_ptr()))
    _lib.check(rc, "pp_decode_f32")
    return out


def get_heatmap_expected_value(heatmaps, sigmas, parzen_size: float = 0.1,
                               return_heatmap: bool = False, backend: str = "hip"):
    """Reference heatmap.py:291-395.  ``heatmaps`` (K,H,W) or (B,K,H,W), ndarray
    or tensor; returns numpy locs (K,2)/(B,K,2) f32 and vals (K,)/(B,K) f32
    (+ the convolved maps with ``return_heatmap``).  The reference raises for
    B > 1 (heatmap.py:362-364); here the batched result is defined as the stack
    of per-crop results.  ``backend`` is accepted for signature compatibility:
    "scipy"/"torch" select CPU libraries in the reference, every value runs
    the HIP kernel here."""
    if not isinstance(heatmaps, (np.ndarray, torch.Tensor)):
        raise AssertionError("heatmaps should be numpy.ndarray or torch.Tensor")
    if heatmaps.ndim not in (3, 4):
        raise AssertionError(f"Invalid shape {tuple(heatmaps.shape)}")
    if not (0.0 <= parzen_size <= 1.0):
        raise AssertionError(f"Invalid parzen_size {parzen_size}")
    _lib.require_device()
    t = torch.from_numpy(np.ascontiguousarray(heatmaps)) if isinstance(heatmaps, np.ndarray) else heatmaps
    squeeze = t.ndim == 3
    if squeeze:
        t = t[None]
    if not t.is_cuda:
        t = t.cuda(non_blocking=True)
    out = decode_on_device(t, sigmas, want_conv=return_heatmap)
    locs, vals = out["locs"].cpu().numpy(), out["scores"].cpu().numpy()
    conv = out["conv"].cpu().numpy() if return_heatmap else None
    if squeeze or t.shape[0] == 1 and heatmaps.ndim == 4:
        # reference reshapes B == 1 results to (K,2)/(K,)/(K,H,W) (heatmap.py:387-390)
        locs, vals = locs[0], vals[0]
        conv = conv[0] if conv is not None else None
    if return_heatmap:
        return locs, vals, conv
    return locs, vals
