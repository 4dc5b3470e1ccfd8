"""Evaluation metrics of the reference as BATCHED operations (second half).

The reference evaluates one instance at a time on the host: ``compute_oks`` (loss.py:715-764) for one (annotation,
detection) pair, ``keypoint_pck_accuracy`` (loss.py:825-866) through a Python loop over keypoints around
heatmap.py:55-111, ``pose_pck_accuracy`` (loss.py:767-822) after a per-map arg-max (heatmap.py:13-52).  Here:

* ``oks_batch`` -- object keypoint similarity of N pairs at once, array-in / array-out; ``compute_oks`` is the
  one-pair view of it with the reference's dict interface.  K numbers per instance: float64 array arithmetic on the host
  (the float64 ``exp`` of a GPU math library is not the host libm's bit for bit, and this metric is pinned bit for bit).
* ``pck_counts`` / ``keypoint_pck_accuracy`` -- one HIP pass over all N x K pairs (``pp_pck_counts``): per-keypoint hit
  and valid COUNTS come back as 2 K integers; the accuracies are their quotients.
* ``pose_pck_accuracy`` -- heatmaps in, accuracy out: both arg-max passes and the counting run on the device, only
  the counts leave it.

Results are identical to the reference's (tests/golden/metrics.npz, minted from the imported reference).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from . import _lib
from .heatmap import get_heatmap_expected_value


# ----------------------------------------------------------------------------------------------- arg-max
def get_heatmap_maximum(heatmaps) -> Tuple[np.ndarray, np.ndarray]:
    """heatmap.py:13-52: (K,H,W) or (B,K,H,W) -> locs (..,K,2) f32 (x, y; -1 where the maximum is <= 0), vals (..,K).
    numpy or device tensor in, numpy out; the arg-max runs on the GPU (first maximum in row-major order, like
    np.argmax)."""
    locs, vals, squeeze = _argmax_device(heatmaps)
    locs, vals = locs.cpu().numpy(), vals.cpu().numpy()
    return (locs[0], vals[0]) if squeeze else (locs, vals)


def _argmax_device(heatmaps):
    _lib.require_device()
    t = torch.from_numpy(np.ascontiguousarray(heatmaps)) if isinstance(heatmaps, np.ndarray) else heatmaps
    assert t.ndim == 3 or t.ndim == 4, f"Invalid shape {tuple(t.shape)}"
    squeeze = t.ndim == 3
    if squeeze:
        t = t[None]
    B, K, H, W = t.shape
    t = t.to(device="cuda", dtype=torch.float32).contiguous()
    locs = torch.empty((B, K, 2), dtype=torch.float32, device="cuda")
    vals = torch.empty((B, K), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().pp_heatmap_argmax(_lib.ptr(t), B * K, H, W, _lib.ptr(locs), _lib.ptr(vals),
                                            _lib.stream_ptr()), "pp_heatmap_argmax")
    return locs, vals, squeeze


# ----------------------------------------------------------------------------------------------- PCK
def _as_device(x, dtype):
    t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
    return t.to(device="cuda", dtype=dtype).contiguous()


def pck_counts(pred, gt, mask, thr, norm_factor, return_distances: bool = False):
    """Per-keypoint (hits, valid) over a batch: ``pred`` / ``gt`` (N,K,2), ``mask`` (N,K) bool, ``norm_factor`` (N,2).

    A pair counts as valid when its mask is set and its instance's normalisation factor holds no exact zero; it is a
    hit when the float32 normalised distance is below ``thr`` (heatmap.py:55-111 semantics, incl. "factor <= 0 -> 1e6").
    numpy arrays or device tensors; coordinates keep their dtype (float32 locations are subtracted in float32 as numpy
    does; anything else is handled in float64).  Returns two int64 arrays of length K (and the (K,N) float32 distance
    matrix, -1 = not valid, when asked)."""
    _lib.require_device()
    N, K = int(pred.shape[0]), int(pred.shape[1])
    is_f32 = (pred.dtype in (np.float32, torch.float32)) and (gt.dtype in (np.float32, torch.float32))
    cdt = torch.float32 if is_f32 else torch.float64
    p, g = _as_device(pred, cdt), _as_device(gt, cdt)
    m = _as_device(mask, torch.bool).view(torch.uint8)
    nf = np.asarray(norm_factor.cpu() if isinstance(norm_factor, torch.Tensor) else norm_factor)
    skip = (nf == 0).any(axis=1)
    nf64 = np.where(nf <= 0, 1e6, nf).astype(np.float64)
    # `array_f32 < python_float` compares in float32 under numpy 2 (NEP 50: the Python scalar is weak); a numpy
    # float64 scalar or array threshold would promote the comparison to float64
    thr_eff = float(np.float32(thr)) if isinstance(thr, (float, int)) and not isinstance(thr, np.generic) else float(thr)
    counts = torch.empty((2, K), dtype=torch.int32, device="cuda")
    dist = torch.empty((K, N), dtype=torch.float32, device="cuda") if return_distances else None
    # every operand stays referenced until the results are back: a temporary freed right after its pointer was taken
    # goes back to the caching allocator and is handed to the next allocation before the kernel has run
    nf_dev = _as_device(nf64, torch.float64)
    skip_dev = _as_device(skip, torch.bool).view(torch.uint8)
    _lib.check(_lib.lib().pp_pck_counts(_lib.ptr(p), _lib.ptr(g), 0 if is_f32 else 1, _lib.ptr(m), _lib.ptr(nf_dev),
                                        _lib.ptr(skip_dev), thr_eff, N, K, _lib.ptr(counts), _lib.ptr(dist),
                                        _lib.stream_ptr()), "pp_pck_counts")
    c = counts.cpu().numpy().astype(np.int64)
    del nf_dev, skip_dev
    return (c[0], c[1], dist.cpu().numpy()) if return_distances else (c[0], c[1])


def _accuracy_from_counts(hits: np.ndarray, valid: np.ndarray) -> tuple:
    acc = np.full(hits.shape, -1.0)
    seen = valid > 0
    acc[seen] = hits[seen] / valid[seen]
    cnt = int(seen.sum())
    return acc, (acc[seen].mean() if cnt else 0.0), cnt


def keypoint_pck_accuracy(pred, gt, mask, thr, norm_factor) -> tuple:
    """loss.py:825-866: (per-keypoint PCK with -1 for keypoints nobody has, their mean over the others, how many others).
    Like the reference (heatmap.py:82) a numpy ``norm_factor`` comes back with its non-positive entries set to 1e6."""
    hits, valid = pck_counts(pred, gt, mask, thr, norm_factor)
    if isinstance(norm_factor, np.ndarray):
        norm_factor[norm_factor <= 0] = 1e6
    return _accuracy_from_counts(hits, valid)


def _default_normalize(N, H, W):
    return np.tile(np.array([[H, W]]), (N, 1))


def pose_pck_accuracy(output, target, mask: np.ndarray, thr: float = 0.05, normalize: np.ndarray | None = None,
                      method: str = "argmax") -> tuple:
    """loss.py:767-822: PCK between the peak locations of two heatmap batches (N,K,H,W) (numpy or device tensors).
    ``method`` 'argmax' locates peaks with get_heatmap_maximum; 'expected' is the branch in which the reference calls
    get_heatmap_expected_value WITHOUT sigmas (loss.py:820-821: a TypeError there) -- ``pose_pck_accuracy_expected``
    is that branch made callable."""
    how = str(method).lower()
    if how not in ("argmax", "expected"):
        raise ValueError(f"Invalid method: {method}")
    if how == "expected":
        raise TypeError("get_heatmap_expected_value() missing 1 required positional argument: 'sigmas' "
                        "(reference loss.py:820 calls it without sigmas; use pose_pck_accuracy_expected)")
    N, K, H, W = output.shape
    if K == 0:
        return None, 0, 0
    p, _, _ = _argmax_device(output)
    g, _, _ = _argmax_device(target)
    hits, valid = pck_counts(p, g, mask, thr, _default_normalize(N, H, W) if normalize is None else normalize)
    return _accuracy_from_counts(hits, valid)


def pose_pck_accuracy_expected(output, target, mask: np.ndarray, sigmas, thr: float = 0.05,
                               normalize: np.ndarray | None = None) -> tuple:
    """The 'expected' method of pose_pck_accuracy made callable: both batches decoded by the fused expected-value
    kernel (per-crop semantics of heatmap.py:291-395) with the given sigmas."""
    N, K, H, W = output.shape
    if K == 0:
        return None, 0, 0
    p, _ = get_heatmap_expected_value(output, sigmas)
    g, _ = get_heatmap_expected_value(target, sigmas)
    hits, valid = pck_counts(np.asarray(p), np.asarray(g), mask, thr,
                             _default_normalize(N, H, W) if normalize is None else normalize)
    return _accuracy_from_counts(hits, valid)


# ----------------------------------------------------------------------------------------------- OKS
def _rowsum_like_numpy(values: np.ndarray, keep: np.ndarray) -> np.ndarray:
    """sum(values[n, keep[n]]) for every row n, each row summed exactly as ``np.sum`` sums the compacted 1-D array
    (numpy's pairwise scheme assigns elements to partial sums by POSITION, so zero-filling the dropped entries would
    change the association): rows are grouped by how many entries they keep, each group compacted into a dense
    (rows, count) block and reduced along its contiguous axis."""
    out = np.zeros(values.shape[0], dtype=np.float64)
    count = keep.sum(axis=1)
    order = np.argsort(~keep, axis=1, kind="stable")        # kept columns first, original order preserved
    packed = np.take_along_axis(values, order, axis=1)
    for c in np.unique(count):
        if c == 0:
            continue
        rows = np.nonzero(count == c)[0]
        out[rows] = np.ascontiguousarray(packed[rows, :c]).sum(axis=1)
    return out


def oks_batch(gt_keypoints, dt_keypoints, bboxes, areas, sigmas, use_area: bool = True, per_kpt: bool = False):
    """COCO object keypoint similarity of N (annotation, detection) pairs (loss.py:715-764 for every pair at once).

    gt_keypoints / dt_keypoints (N,K,3) = (x, y, visibility); bboxes (N,4) = (x, y, w, h); areas (N,) (used when
    ``use_area``, else 0.53 * w * h).  Pairs whose annotation has no visible keypoint are scored by the detection's
    distance to the box grown by its own size on every side, over all K keypoints; the others by the keypoint
    distances over the visible keypoints.  Returns (N,) similarities, or (N,K) per-keypoint terms (0 at invisible
    keypoints of annotated instances) with ``per_kpt``."""
    G = np.asarray(gt_keypoints, dtype=np.float64)
    D = np.asarray(dt_keypoints, dtype=np.float64)
    box = np.asarray(bboxes, dtype=np.float64).reshape(-1, 4)
    N, K = G.shape[0], G.shape[1]
    variance = (np.asarray(sigmas) * 2) ** 2
    seen = G[..., 2] > 0
    annotated = seen.any(axis=1)
    # distance to the annotation where there is one ...
    delta = D[..., :2] - G[..., :2]
    # ... and to the grown box where there is none
    lo, hi = box[:, None, :2] - box[:, None, 2:], box[:, None, :2] + box[:, None, 2:] * 2
    outside = np.maximum(0.0, lo - D[..., :2]) + np.maximum(0.0, D[..., :2] - hi)
    delta = np.where(annotated[:, None, None], delta, outside)
    size = (np.asarray(areas, dtype=np.float64) if use_area else box[:, 3] * box[:, 2] * 0.53) + np.spacing(1)
    similarity = np.exp(-((delta[..., 0] ** 2 + delta[..., 1] ** 2) / variance / size[:, None] / 2))
    if per_kpt:
        similarity[annotated[:, None] & ~seen] = 0
        return similarity
    counted = np.where(annotated[:, None], seen, True)
    return _rowsum_like_numpy(similarity, counted) / counted.sum(axis=1)


def compute_oks(gt, dt, sigmas: np.ndarray, use_area=True, per_kpt=False):
    """loss.py:715-764, the reference's one-pair interface: ``gt`` = {"keypoints": 3K numbers, "bbox": [x, y, w, h],
    "area": a}, ``dt`` = {"keypoints": 3K numbers} -> similarity (or the K per-keypoint terms)."""
    k = len(sigmas)
    out = oks_batch(np.asarray(gt["keypoints"]).reshape(1, k, 3), np.asarray(dt["keypoints"]).reshape(1, k, 3),
                    [gt["bbox"]], [gt["area"]] if use_area else [0.0], sigmas, use_area=use_area, per_kpt=per_kpt)
    return out[0]
