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
    from probpose_pytorch_amd import metrics
    sigmas, gts, dts = _inputs()[:3]
    for i, (g, d) in enumerate(zip(gts, dts)):
        assert metrics.compute_oks(g, d, sigmas, use_area=True) == gold[f"oks_area_{i}"]
        assert metrics.compute_oks(g, d, sigmas, use_area=False) == gold[f"oks_bbox_{i}"]
        np.testing.assert_array_equal(metrics.compute_oks(g, d, sigmas, use_area=True, per_kpt=True),
                                      gold[f"oks_perkpt_{i}"])


def test_keypoint_pck_accuracy_matches_reference(gold):
    from probpose_pytorch_amd import metrics
    _, _, _, _, _, mask, pred, gt = _inputs()
    norm = np.tile(np.array([[48.0, 64.0]]), (pred.shape[0], 1))
    norm[2] = 0.0
    acc, avg, cnt = metrics.keypoint_pck_accuracy(pred, gt, mask, 0.05, norm.copy())
    np.testing.assert_array_equal(acc, gold["kpck_acc"])
    assert avg == gold["kpck_avg"] and cnt == gold["kpck_cnt"]


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
