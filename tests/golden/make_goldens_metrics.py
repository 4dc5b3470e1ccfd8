"""Mint tests/golden/metrics.npz by running the reference's own metric functions in this container
(reference loss.py:715-866, heatmap.py:13-111): compute_oks, keypoint_pck_accuracy, pose_pck_accuracy('argmax'),
get_heatmap_maximum.  cv2 is registered as an empty stub exactly as in make_goldens.py (imported by the reference,
unused by these functions).  Run once, here:  cd /tmp && python /root/repo/tests/golden/make_goldens_metrics.py
Only outputs (and seeds) are stored; the tests regenerate the inputs from the seeds."""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

import numpy as np  # noqa: E402

sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, REF)
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
import probpose  # noqa: E402

assert all(os.path.abspath(p).startswith(REF) for p in probpose.__path__)
from probpose.heatmap import get_heatmap_maximum  # noqa: E402
from probpose.loss import compute_oks, keypoint_pck_accuracy, pose_pck_accuracy  # noqa: E402


def metric_inputs(seed: int):
    """Seeded inputs shared with tests/test_metrics.py (kept identical there)."""
    rng = np.random.default_rng(seed)
    K = 17
    sigmas = np.array([.026, .025, .025, .035, .035, .079, .079, .072, .072, .062, .062, .107, .107, .087, .087,
                       .089, .089])
    gts, dts = [], []
    for case in range(4):
        kp = rng.uniform(0, 200, (K, 3))
        kp[:, 2] = rng.integers(0, 3, K)
        if case == 2:
            kp[:, 2] = 0                      # no visible keypoint: the bounding-box distance branch
        bbox = [20.0 + case, 30.0, 120.0, 160.0]
        gts.append(dict(keypoints=kp.reshape(-1).tolist(), bbox=bbox, area=float(bbox[2] * bbox[3] * 0.6)))
        d = kp.copy()
        d[:, :2] += rng.normal(0, 6.0, (K, 2))
        dts.append(dict(keypoints=d.reshape(-1).tolist()))
    N, H, W = 6, 64, 48
    hm_out = rng.random((N, K, H, W), dtype=np.float32)
    hm_tgt = hm_out.copy()
    # move a third of the target peaks
    for n in range(N):
        for k in range(K):
            y, x = rng.integers(0, H), rng.integers(0, W)
            hm_out[n, k, y, x] = 2.0
            if rng.random() < 0.6:
                hm_tgt[n, k, y, x] = 2.0
            else:
                hm_tgt[n, k, min(H - 1, y + rng.integers(0, 6)), min(W - 1, x + rng.integers(0, 6))] = 2.0
    hm_out[1, 3] = 0.0                         # a dead map: locs -1
    mask = rng.random((N, K)) > 0.2
    pred = rng.uniform(0, 48, (N, K, 2)).astype(np.float32)
    gt = pred + rng.normal(0, 2.0, (N, K, 2)).astype(np.float32)
    return sigmas, gts, dts, hm_out, hm_tgt, mask, pred, gt


def main():
    sigmas, gts, dts, hm_out, hm_tgt, mask, pred, gt = metric_inputs(2025)
    out = {}
    for i, (g, d) in enumerate(zip(gts, dts)):
        out[f"oks_area_{i}"] = np.float64(compute_oks(g, d, sigmas, use_area=True))
        out[f"oks_bbox_{i}"] = np.float64(compute_oks(g, d, sigmas, use_area=False))
        out[f"oks_perkpt_{i}"] = compute_oks(g, d, sigmas, use_area=True, per_kpt=True)
    locs, vals = get_heatmap_maximum(hm_out)
    out["max_locs"], out["max_vals"] = locs, vals
    l3, v3 = get_heatmap_maximum(hm_out[0])
    out["max_locs3"], out["max_vals3"] = l3, v3
    acc, avg, cnt = pose_pck_accuracy(hm_out, hm_tgt, mask, thr=0.05)
    out["pck_acc"], out["pck_avg"], out["pck_cnt"] = acc, np.float64(avg), np.int64(cnt)
    norm = np.tile(np.array([[48.0, 64.0]]), (pred.shape[0], 1))
    norm[2] = 0.0                              # an instance with a zero normaliser is masked out (heatmap.py:80-81)
    acc2, avg2, cnt2 = keypoint_pck_accuracy(pred, gt, mask, 0.05, norm.copy())
    out["kpck_acc"], out["kpck_avg"], out["kpck_cnt"] = acc2, np.float64(avg2), np.int64(cnt2)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), seed=2025, **out)
    print("metrics.npz:", {k: (np.asarray(v).shape, np.asarray(v).dtype) for k, v in out.items()})


if __name__ == "__main__":
    main()
