"""Evaluation-side components of SURVEY.md section 8f:
  * rank 2 -- ArgMaxProbMap.decode = raw arg-max + DARK-UDP (reference codec.py:284-375, 515-543; heatmap.py:13-52):
    cv2 is not importable, so the blur is restated from OpenCV's published algorithm and PARITY IS UNPINNED for the
    decoder as a whole; its cv2-free part (get_heatmap_maximum) is pinned by a golden minted from the reference;
  * rank 3 (second half) -- compute_oks / pose_pck_accuracy / keypoint_pck_accuracy (loss.py:715-866): pinned
    bit-for-bit by tests/golden/metrics.npz, minted by tests/golden/make_goldens_metrics.py from the imported reference.
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import probpose_oracle as orc
from tests.helpers import GOLDEN

_spec = importlib.util.spec_from_file_location("mk_metrics", os.path.join(GOLDEN, "make_goldens_metrics.py"))


def _inputs():
    # the golden script's seeded input generator, loaded WITHOUT running its reference imports
    src = open(os.path.join(GOLDEN, "make_goldens_metrics.py")).read()
    start = src.index("def metric_inputs")
    end = src.index("def main():")
    ns = {"np": np}
    exec(src[start:end], ns)
    return ns["metric_inputs"](2025)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "metrics.npz"))


def test_compute_oks_matches_reference(gold):
    """The product's batched OKS (one-pair view) and the oracle's per-instance restatement, both against the goldens."""
    from probpose_pytorch_amd import metrics
    sigmas, gts, dts = _inputs()[:3]
    for impl in (metrics.compute_oks, orc.compute_oks):
        for i, (g, d) in enumerate(zip(gts, dts)):
            assert impl(g, d, sigmas, use_area=True) == gold[f"oks_area_{i}"]
            assert impl(g, d, sigmas, use_area=False) == gold[f"oks_bbox_{i}"]
            np.testing.assert_array_equal(impl(g, d, sigmas, use_area=True, per_kpt=True), gold[f"oks_perkpt_{i}"])


def test_oks_batch_equals_per_instance(gold):
    """All pairs in ONE call (mixed visibility counts, an un-annotated instance) == the goldens of the single calls;
    larger random batches (K = 17 and K = 133 > numpy's 128-element pairwise block) == the oracle's per-instance loop."""
    from probpose_pytorch_amd import metrics
    sigmas, gts, dts = _inputs()[:3]
    K = len(sigmas)
    G = np.array([g["keypoints"] for g in gts]).reshape(-1, K, 3)
    D = np.array([d["keypoints"] for d in dts]).reshape(-1, K, 3)
    box = np.array([g["bbox"] for g in gts])
    area = np.array([g["area"] for g in gts])
    for use_area, key in ((True, "oks_area_"), (False, "oks_bbox_")):
        got = metrics.oks_batch(G, D, box, area, sigmas, use_area=use_area)
        np.testing.assert_array_equal(got, np.array([gold[f"{key}{i}"] for i in range(len(gts))]))
    np.testing.assert_array_equal(metrics.oks_batch(G, D, box, area, sigmas, per_kpt=True),
                                  np.stack([gold[f"oks_perkpt_{i}"] for i in range(len(gts))]))
    rng = np.random.default_rng(11)
    for K2, N in ((17, 300), (133, 120)):
        sig = rng.uniform(0.02, 0.11, K2)
        G = rng.uniform(0, 300, (N, K2, 3))
        G[..., 2] = rng.integers(0, 3, (N, K2))
        G[rng.random(N) < 0.15, :, 2] = 0                         # un-annotated instances: box-distance branch
        D = G + rng.normal(0, 8.0, G.shape)
        box = np.concatenate([rng.uniform(0, 100, (N, 2)), rng.uniform(20, 200, (N, 2))], axis=1)
        area = box[:, 2] * box[:, 3] * rng.uniform(0.3, 0.9, N)
        for use_area in (True, False):
            want = np.array([orc.compute_oks(dict(keypoints=G[n].reshape(-1), bbox=list(box[n]), area=area[n]),
                                             dict(keypoints=D[n].reshape(-1)), sig, use_area=use_area) for n in range(N)])
            np.testing.assert_array_equal(metrics.oks_batch(G, D, box, area, sig, use_area=use_area), want)


def test_oracle_pck_matches_reference(gold):
    _, _, _, hm_out, hm_tgt, mask, pred, gt = _inputs()
    norm = np.tile(np.array([[48.0, 64.0]]), (pred.shape[0], 1))
    norm[2] = 0.0
    acc, avg, cnt = orc.keypoint_pck_accuracy(pred, gt, mask, 0.05, norm.copy())
    np.testing.assert_array_equal(acc, gold["kpck_acc"])
    assert avg == gold["kpck_avg"] and cnt == gold["kpck_cnt"]
    acc, avg, cnt = orc.pose_pck_accuracy(hm_out, hm_tgt, mask, thr=0.05)
    np.testing.assert_array_equal(acc, gold["pck_acc"])
    assert avg == gold["pck_avg"] and cnt == gold["pck_cnt"]


@pytest.mark.gpu
def test_keypoint_pck_accuracy_matches_reference(built_lib, gold):
    """pp_pck_counts (one pass over all N x K pairs) against the reference goldens and, on a large ragged batch, against
    the oracle's per-instance loop: identical counts, accuracies and distance matrix."""
    from probpose_pytorch_amd import metrics
    _, _, _, _, _, mask, pred, gt = _inputs()
    norm = np.tile(np.array([[48.0, 64.0]]), (pred.shape[0], 1))
    norm[2] = 0.0
    nf = norm.copy()
    acc, avg, cnt = metrics.keypoint_pck_accuracy(pred, gt, mask, 0.05, nf)
    np.testing.assert_array_equal(acc, gold["kpck_acc"])
    assert avg == gold["kpck_avg"] and cnt == gold["kpck_cnt"]
    assert (nf[2] == 1e6).all()                                   # the reference's in-place edit (heatmap.py:82)
    rng = np.random.default_rng(5)
    for N, K, dt in ((4097, 17, np.float32), (513, 133, np.float32), (257, 17, np.float64)):
        p = rng.uniform(0, 96, (N, K, 2)).astype(dt)
        g = (p + rng.normal(0, 3.0, (N, K, 2))).astype(dt)
        m = rng.random((N, K)) > 0.3
        m[:, 3] = False                                           # a keypoint nobody has: accuracy -1
        nrm = rng.uniform(20, 100, (N, 2))
        nrm[rng.random(N) < 0.05] = 0.0                           # instances dropped altogether
        nrm[rng.random(N) < 0.05, 1] = -3.0                       # negative factor -> 1e6
        for thr in (0.05, np.float64(0.05), 0.2):
            want = orc.keypoint_pck_accuracy(p, g, m, thr, nrm.copy())
            got = metrics.keypoint_pck_accuracy(p, g, m, thr, nrm.copy())
            np.testing.assert_array_equal(got[0], want[0])
            assert got[1] == want[1] and got[2] == want[2]
        hits, valid, dist = metrics.pck_counts(p, g, m, 0.05, nrm.copy(), return_distances=True)
        np.testing.assert_array_equal(dist, orc.normalized_distances(p, g, m, nrm.copy()))
        assert dist.dtype == np.float32 and valid[3] == 0
    # empty batch / device tensors
    h0, v0 = metrics.pck_counts(np.zeros((0, 17, 2), np.float32), np.zeros((0, 17, 2), np.float32),
                                np.zeros((0, 17), bool), 0.05, np.zeros((0, 2)))
    assert h0.sum() == 0 and v0.sum() == 0
    got_t = metrics.keypoint_pck_accuracy(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda(), mask, 0.05, norm.copy())
    np.testing.assert_array_equal(got_t[0], gold["kpck_acc"])


@pytest.mark.gpu
def test_heatmap_maximum_any_size(built_lib):
    """pp_heatmap_argmax: maps that exceed any LDS budget (256x256, the reference test's own shape), one-pixel-wide and
    one-pixel-high maps, odd sizes (scalar path), ties (first index), NaN (counts as the maximum), dead maps (-1)."""
    from probpose_pytorch_amd import metrics
    rng = np.random.default_rng(9)
    for shape in ((2, 3, 256, 256), (1, 5, 1, 37), (1, 4, 41, 1), (3, 2, 33, 27), (17, 64, 48), (2, 133, 96, 72)):
        hm = rng.random(shape, dtype=np.float32)
        flat = hm.reshape(-1, shape[-2], shape[-1])
        flat[0, 0, 0] = flat[0].max()                              # tie with a later pixel: the first one wins
        if flat.shape[0] > 1:
            flat[1] = -flat[1]                                     # all <= 0: dead map
        if flat.shape[0] > 2:
            flat[2, shape[-2] // 2, shape[-1] // 2] = np.nan
        locs, vals = metrics.get_heatmap_maximum(hm)
        wl, wv = orc.get_heatmap_maximum(hm)
        np.testing.assert_array_equal(locs, wl)
        np.testing.assert_array_equal(vals, wv)
        assert locs.shape == shape[:-2] + (2,) and locs.dtype == np.float32


def test_oracle_heatmap_maximum_matches_reference(gold):
    hm_out = _inputs()[3]
    locs, vals = orc.get_heatmap_maximum(hm_out)
    np.testing.assert_array_equal(locs, gold["max_locs"])
    np.testing.assert_array_equal(vals, gold["max_vals"])
    l3, v3 = orc.get_heatmap_maximum(hm_out[0])
    np.testing.assert_array_equal(l3, gold["max_locs3"])
    assert (gold["max_locs"][1, 3] == -1).all()              # the dead map


def test_oracle_dark_blur_properties():
    """The restated blur: a symmetric float32 kernel summing to 1, zero-padded (a constant map loses mass only at the
    border), rescaled to the original maximum; DARK moves an exact Gaussian blob's arg-max onto its true centre."""
    k = orc.gaussian_kernel_f32(11)
    assert k.dtype == np.float32 and abs(float(k.astype(np.float64).sum()) - 1) < 1e-6
    np.testing.assert_array_equal(k, k[::-1])
    hm = np.zeros((64, 48), np.float32)
    hm[20, 30] = 0.8
    out = orc.gaussian_blur_zero_padded(hm, 11)
    assert np.unravel_index(out.argmax(), out.shape) == (20, 30) and abs(float(out.max()) - 0.8) < 1e-6
    # sub-pixel recovery: Gaussian blob centred off-grid, sigma 2 (log of a Gaussian is a parabola)
    yy, xx = np.mgrid[0:64, 0:48].astype(np.float64)
    cx, cy = 21.3, 40.6
    blob = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 2.0 ** 2)).astype(np.float32)[None]
    kp, sc = orc.dark_udp_decode(blob, 11, (192, 256), (48, 64))
    got = kp[0, 0] / [192, 256] * [47, 63]
    assert abs(got[0] - cx) < 0.02 and abs(got[1] - cy) < 0.02
    assert sc.shape == (1, 1) and abs(float(sc[0, 0]) - float(blob.max())) == 0


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_get_heatmap_maximum_and_pck_on_gpu(built_lib, gold):
    from probpose_pytorch_amd import metrics
    _, _, _, hm_out, hm_tgt, mask, _, _ = _inputs()
    locs, vals = metrics.get_heatmap_maximum(hm_out)
    np.testing.assert_array_equal(locs, gold["max_locs"])
    np.testing.assert_array_equal(vals, gold["max_vals"])
    l3, v3 = metrics.get_heatmap_maximum(torch.from_numpy(hm_out[0]).cuda())
    np.testing.assert_array_equal(l3, gold["max_locs3"])
    np.testing.assert_array_equal(v3, gold["max_vals3"])
    acc, avg, cnt = metrics.pose_pck_accuracy(hm_out, hm_tgt, mask, thr=0.05)
    np.testing.assert_array_equal(acc, gold["pck_acc"])
    assert avg == gold["pck_avg"] and cnt == gold["pck_cnt"]
    with pytest.raises(TypeError):                 # the reference's 'expected' branch omits sigmas (loss.py:820)
        metrics.pose_pck_accuracy(hm_out, hm_tgt, mask, method="expected")
    acc_e, avg_e, cnt_e = metrics.pose_pck_accuracy_expected(hm_out[:2], hm_out[:2], mask[:2], orc.COCO17_SIGMAS)
    assert avg_e == 1.0 and cnt_e > 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,H,W,ks,in_size", [(4, 17, 64, 48, 11, (192, 256)), (2, 133, 96, 72, 11, (288, 384)),
                                                (2, 20, 96, 96, 17, (384, 384)), (3, 5, 33, 27, 5, (108, 132))])
def test_dark_udp_decode_matches_restatement(built_lib, B, K, H, W, ks, in_size):
    """pp_dark_decode_f32 vs the oracle's restatement of ArgMaxProbMap.decode, per crop (parity unpinned: cv2)."""
    from probpose_pytorch_amd import ArgMaxProbMap
    hm = orc.synthetic_heatmaps(B, K, H, W, seed=B * 100 + K, kind="peaked")
    hm[0, 1] = 0.0                                            # dead map -> (-1, -1), un-refined
    hm[B - 1, 2] = np.random.default_rng(3).random((H, W), dtype=np.float32)   # noise: ill-conditioned Hessians
    codec = ArgMaxProbMap(in_size, (W, H), blur_kernel_size=ks)
    kp, sc = codec.decode(hm)
    assert kp.shape == (B, K, 2) and kp.dtype == np.float64 and sc.shape == (B, K) and sc.dtype == np.float32
    for b in range(B):
        want_kp, want_sc = orc.dark_udp_decode(hm[b], ks, in_size, (W, H))
        np.testing.assert_array_equal(sc[b], want_sc[0])
        d = np.abs(kp[b] - want_kp[0]).max(-1)
        # logf / np.log differ by <= 1 ulp; the Hessian step amplifies it by 1 / |dxx|: negligible on peaked maps,
        # visible on the flat noise map (b = B - 1, k = 2), which gets its own, looser bound
        noise = np.zeros(K, bool)
        if b == B - 1:
            noise[2] = True
        assert (d[~noise] <= 1e-4).all() and np.median(d) <= 1e-5, (d[~noise].max(), np.median(d))
        assert (d[noise] <= 5e-2).all(), d[noise]
    np.testing.assert_allclose(kp[0, 1], np.array([-1.0, -1.0]) / [W - 1, H - 1] * in_size)
    # single (K,H,W) input keeps the reference's leading instance axis
    kp1, sc1 = codec.decode(hm[0])
    assert kp1.shape == (1, K, 2) and sc1.shape == (1, K)
    np.testing.assert_array_equal(kp1[0], kp[0])


@pytest.mark.gpu
def test_codec_with_argmax_probmap(built_lib):
    """Codec(ArgMaxProbMap(...)).decode keeps the 5-tuple contract of codec.py:249-263."""
    from probpose_pytorch_amd import ArgMaxProbMap, Codec
    B, K, H, W = 3, 17, 64, 48
    hm = orc.synthetic_heatmaps(B, K, H, W, seed=11)
    rng = np.random.default_rng(1)
    aux = [rng.random((B, K, 1, 1), dtype=np.float32) for _ in range(4)]
    codec = Codec(ArgMaxProbMap((192, 256), (W, H), sigmas=orc.COCO17_SIGMAS))
    (kp, sc), prob, vis, oks, err = codec.decode(tuple(torch.from_numpy(a).cuda() for a in (hm, *aux)))
    assert kp.shape == (B, K, 2) and prob.shape == (B, 1, K)
    np.testing.assert_array_equal(prob, aux[0].reshape(B, 1, K))
    np.testing.assert_allclose(err, aux[3].reshape(B, 1, K).astype(np.float64) / np.sqrt(H * H + W * W), rtol=1e-15)
