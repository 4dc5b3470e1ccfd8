"""Mirror of the evaluation metrics of the reference (second half):
``compute_oks`` (loss.py:715-764), ``pose_pck_accuracy`` (loss.py:767-822), ``keypoint_pck_accuracy``
(loss.py:825-866) and the helpers they use from heatmap.py (``get_heatmap_maximum`` :13-52, ``_calc_distances``
:55-89, ``_distance_acc`` :92-111).

The per-instance OKS / PCK arithmetic is K numbers per instance and stays on the host in numpy, exactly as the
reference evaluates it (pinned bit-for-bit by tests/golden/metrics.npz, minted from the imported reference).  What
the reference does per heatmap on the host -- the arg-max of every map (``get_heatmap_maximum``) and the
expected-value decode (``get_heatmap_expected_value``) -- runs in the HIP decode kernels on the whole batch, and only
B*K*2 coordinates come back.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from . import _lib
from .heatmap import get_heatmap_expected_value


def get_heatmap_maximum(heatmaps) -> Tuple[np.ndarray, np.ndarray]:
    """heatmap.py:13-52: (K,H,W) or (B,K,H,W) -> locs (..,K,2) f32 (x, y; -1 where the maximum is <= 0), vals (..,K).
    numpy or device tensor in, numpy out; the arg-max runs on the GPU (first maximum in row-major order, like
    np.argmax)."""
    _lib.require_device()
    t = torch.from_numpy(np.ascontiguousarray(heatmaps)) if isinstance(heatmaps, np.ndarray) else heatmaps
    assert t.ndim == 3 or t.ndim == 4, f"Invalid shape {tuple(t.shape)}"
    squeeze = t.ndim == 3
    if squeeze:
        t = t[None]
    from .codec import ArgMaxProbMap
    B, K, H, W = t.shape
    out = ArgMaxProbMap((W, H), (W, H), blur_kernel_size=1).decode_device(t.cuda() if not t.is_cuda else t)
    locs, vals = out["locs"].cpu().numpy(), out["scores"].cpu().numpy()
    return (locs[0], vals[0]) if squeeze else (locs, vals)


def _calc_distances(preds: np.ndarray, gts: np.ndarray, mask: np.ndarray, norm_factor: np.ndarray) -> np.ndarray:
    """heatmap.py:55-89 (note: like the reference, ``norm_factor`` is modified in place where it is <= 0)."""
    N, K, _ = preds.shape
    _mask = mask.copy()
    _mask[np.where((norm_factor == 0).sum(1))[0], :] = False
    distances = np.full((N, K), -1, dtype=np.float32)
    norm_factor[np.where(norm_factor <= 0)] = 1e6
    distances[_mask] = np.linalg.norm(((preds - gts) / norm_factor[:, None, :])[_mask], axis=-1)
    return distances.T


def _distance_acc(distances: np.ndarray, thr: float = 0.5) -> float:
    """heatmap.py:92-111."""
    distance_valid = distances != -1
    num_distance_valid = distance_valid.sum()
    if num_distance_valid > 0:
        return (distances[distance_valid] < thr).sum() / num_distance_valid
    return -1


def keypoint_pck_accuracy(pred: np.ndarray, gt: np.ndarray, mask: np.ndarray, thr, norm_factor: np.ndarray) -> tuple:
    """loss.py:825-866: per-keypoint PCK, their mean over the valid keypoints, and the number of valid keypoints."""
    distances = _calc_distances(pred, gt, mask, norm_factor)
    acc = np.array([_distance_acc(d, thr) for d in distances])
    valid_acc = acc[acc >= 0]
    cnt = len(valid_acc)
    avg_acc = valid_acc.mean() if cnt > 0 else 0.0
    return acc, avg_acc, cnt


def pose_pck_accuracy(output, target, mask: np.ndarray, thr: float = 0.05, normalize: np.ndarray | None = None,
                      method: str = "argmax") -> tuple:
    """loss.py:767-822: PCK between the decoded locations of two heatmap batches (N,K,H,W) (numpy or device tensors).
    ``method`` 'argmax' decodes with get_heatmap_maximum, 'expected' with get_heatmap_expected_value -- which the
    reference calls WITHOUT sigmas (loss.py:820-821: a TypeError there); ``sigmas`` must therefore be supplied through
    ``pose_pck_accuracy_expected`` for that method."""
    method = method.lower()
    if method not in ["argmax", "expected"]:
        raise ValueError(f"Invalid method: {method}")
    N, K, H, W = output.shape
    if K == 0:
        return None, 0, 0
    if normalize is None:
        normalize = np.tile(np.array([[H, W]]), (N, 1))
    if method == "expected":
        raise TypeError("get_heatmap_expected_value() missing 1 required positional argument: 'sigmas' "
                        "(reference loss.py:820 calls it without sigmas; use pose_pck_accuracy_expected)")
    pred, _ = get_heatmap_maximum(output)
    gt, _ = get_heatmap_maximum(target)
    return keypoint_pck_accuracy(pred, gt, mask, thr, normalize)


def pose_pck_accuracy_expected(output, target, mask: np.ndarray, sigmas, thr: float = 0.05,
                               normalize: np.ndarray | None = None) -> tuple:
    """The 'expected' method of pose_pck_accuracy made callable: both batches decoded by the fused expected-value
    kernel (per-crop semantics of heatmap.py:291-395) with the given sigmas."""
    N, K, H, W = output.shape
    if K == 0:
        return None, 0, 0
    if normalize is None:
        normalize = np.tile(np.array([[H, W]]), (N, 1))
    pred, _ = get_heatmap_expected_value(output, sigmas)
    gt, _ = get_heatmap_expected_value(target, sigmas)
    return keypoint_pck_accuracy(np.asarray(pred), np.asarray(gt), mask, thr, normalize)


def compute_oks(gt, dt, sigmas: np.ndarray, use_area=True, per_kpt=False):
    """loss.py:715-764: COCO object keypoint similarity of one detection against one annotation."""
    vars = (sigmas * 2) ** 2
    k = len(sigmas)

    def visibility_condition(x):
        return x > 0

    g = np.array(gt["keypoints"]).reshape(k, 3)
    xg, yg, vg = g[:, 0], g[:, 1], g[:, 2]
    k1 = np.count_nonzero(visibility_condition(vg))
    bb = gt["bbox"]
    x0 = bb[0] - bb[2]
    x1 = bb[0] + bb[2] * 2
    y0 = bb[1] - bb[3]
    y1 = bb[1] + bb[3] * 2
    d = np.array(dt["keypoints"]).reshape((k, 3))
    xd, yd = d[:, 0], d[:, 1]
    if k1 > 0:
        dx = xd - xg
        dy = yd - yg
    else:
        z = np.zeros((k))
        dx = np.max((z, x0 - xd), axis=0) + np.max((z, xd - x1), axis=0)
        dy = np.max((z, y0 - yd), axis=0) + np.max((z, yd - y1), axis=0)
    if use_area:
        e = (dx**2 + dy**2) / vars / (gt["area"] + np.spacing(1)) / 2
    else:
        tmparea = gt["bbox"][3] * gt["bbox"][2] * 0.53
        e = (dx**2 + dy**2) / vars / (tmparea + np.spacing(1)) / 2
    if per_kpt:
        oks = np.exp(-e)
        if k1 > 0:
            oks[~visibility_condition(vg)] = 0
    else:
        if k1 > 0:
            e = e[visibility_condition(vg)]
        oks = np.sum(np.exp(-e)) / e.shape[0]
    return oks
