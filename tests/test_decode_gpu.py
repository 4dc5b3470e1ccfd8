"""GPU parity tests of the fused decode (csrc/pp_decode.hip) through the
Python mirror of the reference API, against (1) the goldens minted from the
reference and (2) the CPU oracle on seeded inputs.

Tolerances: integer work (argmax location, score gather) bit-exact; float32
heatmap-space locs bit-exact (same float32 arithmetic on a bit-identical
convolved map); float64 keypoints within 1e-4 px abs (north_star), observed 0.
"""
import os

import numpy as np
import pytest
import torch

from oracle import probpose_oracle as orc
from tests.helpers import DECODE_FIXTURES, GOLDEN, load_decode_fixture

pytestmark = pytest.mark.gpu

ATOL_KPTS = 1e-4


@pytest.fixture(scope="module")
def pp(built_lib):
    import probpose_pytorch_amd as p
    assert torch.cuda.is_available()
    assert built_lib.pp_device_ok() == 1, "expected a gfx950 device"
    return p


def _dev(arrs):
    return tuple(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs)


@pytest.mark.parametrize("name", DECODE_FIXTURES)
def test_codec_decode_matches_reference_golden(pp, name):
    g, hm, aux = load_decode_fixture(name)
    codec = pp.Codec(pp.ProbMap(tuple(g["in_size"]), (int(g["W"]), int(g["H"])), g["sigmas"]))
    (kpts, scores), prob, vis, oks, err = codec.decode(_dev((hm, *aux)))
    assert kpts.dtype == np.float64 and scores.dtype == np.float32 and err.dtype == np.float64
    assert kpts.shape == g["kpts"].shape and prob.shape == g["prob"].shape
    np.testing.assert_allclose(kpts, g["kpts"], rtol=0, atol=ATOL_KPTS)
    np.testing.assert_array_equal(scores, g["scores"])
    np.testing.assert_array_equal(prob, g["prob"])
    np.testing.assert_array_equal(vis, g["vis"])
    np.testing.assert_array_equal(oks, g["oks"])
    np.testing.assert_allclose(err, g["err"], rtol=1e-15, atol=0)
    # stronger than the contract: expected to be identical
    assert np.abs(kpts - g["kpts"]).max() == 0.0


def test_single_crop_shapes_match_reference(pp):
    """B == 1 is the only case the reference supports: (1,K,2)/(1,K)/(1,1,K)."""
    g, hm, aux = load_decode_fixture("decode_k17_peaked.npz")
    codec = pp.Codec(pp.ProbMap((192, 256), (48, 64), g["sigmas"]))
    out = codec.decode(_dev([a[:1] for a in (hm, *aux)]))
    assert out[0][0].shape == (1, 17, 2) and out[0][1].shape == (1, 17)
    assert all(o.shape == (1, 1, 17) for o in out[1:])
    k2, s2 = codec.decode_heatmap(hm[0])                       # numpy (K,H,W) like loss.py:531
    np.testing.assert_array_equal(k2, out[0][0])
    np.testing.assert_array_equal(s2, out[0][1])
    k3, _ = codec.decode_heatmap(torch.from_numpy(hm[0]).cuda())
    np.testing.assert_array_equal(k3, out[0][0])


def test_convolved_maps_bit_exact(pp):
    g = np.load(os.path.join(GOLDEN, "convmaps_k17.npz"))
    hm = orc.synthetic_heatmaps(1, 17, 64, 48, 4321, "peaked")[0]
    before = hm.copy()
    locs, vals, conv = pp.get_heatmap_expected_value(hm, orc.COCO17_SIGMAS, return_heatmap=True)
    np.testing.assert_array_equal(hm, before)                  # input never mutated
    assert locs.shape == (17, 2) and vals.shape == (17,) and conv.shape == (17, 64, 48)
    np.testing.assert_array_equal(conv, g["conv_scipy"])
    np.testing.assert_array_equal(locs, g["locs"])
    np.testing.assert_array_equal(vals, g["vals"])


def test_large_map_fallback_reference_test_shape(pp):
    """256x256 does not fit in LDS -> three-pass global path (reference tests/test_heatmap.py:6)."""
    g = np.load(os.path.join(GOLDEN, "bigmap_256.npz"))
    rng = np.random.default_rng(2024)
    big = rng.random((20, 256, 256), dtype=np.float32)
    sig = rng.random(20, dtype=np.float32)
    locs, vals, conv = pp.get_heatmap_expected_value(big, sig, return_heatmap=True)
    np.testing.assert_array_equal(conv[:, ::16, ::16], g["conv_sample"])
    np.testing.assert_array_equal(locs, g["locs"])
    np.testing.assert_array_equal(vals, g["vals"])
    # the reference's own assertion: the two back-ends agree to rtol 1e-5
    _, _, ref_conv = orc.heatmap_expected_value(big[:2], sig[:2], "torch", return_heatmap=True)
    np.testing.assert_allclose(conv[:2], ref_conv, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("shape", [(3, 5, 7, 9), (1, 4, 4, 3), (2, 3, 33, 31), (1, 2, 128, 96),
                                   (1, 1, 1, 1), (2, 17, 64, 48)])
@pytest.mark.parametrize("kind", ["uniform", "peaked"])
def test_ragged_shapes_against_oracle(pp, shape, kind):
    """Odd sizes (no float4 path), maps smaller than the kernel radius (multiple
    reflections), the largest map that still fits LDS (128x96)."""
    B, K, H, W = shape
    hm = orc.synthetic_heatmaps(B, K, H, W, seed=H * 1000 + W, kind=kind)
    sig = np.random.default_rng(K).uniform(0.02, 0.2, K)
    got_l, got_v = pp.get_heatmap_expected_value(hm, sig)
    for b in range(B):
        if min(H, W) > 9:
            want_l, want_v = orc.heatmap_expected_value(hm[b], sig, "scipy")
        else:   # the torch back-end cannot pad by more than the size; scipy can
            want_l, want_v = orc.heatmap_expected_value(hm[b], sig, "scipy")
        gl = got_l[b] if B > 1 else got_l
        gv = got_v[b] if B > 1 else got_v
        np.testing.assert_array_equal(gl, want_l)
        np.testing.assert_array_equal(gv, want_v)


def test_edge_maps_zero_flat_ties_negative_nan(pp):
    K, H, W = 8, 64, 48
    hm = np.zeros((K, H, W), np.float32)
    hm[1] = 1.0                                   # saturated: every pixel ties -> index 0
    hm[2, 10, 20] = hm[2, 40, 5] = 0.7            # two equal peaks -> first in row-major order
    hm[3, H - 1, W - 1] = 1.0                     # corner
    hm[4] = -np.random.default_rng(0).random((H, W), dtype=np.float32)   # negative raw input
    hm[5, 30, 0] = 0.9                            # border column
    hm[6, 20:23, 20:23] = 1.0                     # plateau: dxx == 0 -> 1e-6 branch
    hm[7] = np.random.default_rng(1).random((H, W), dtype=np.float32)
    hm[7, 13, 17] = np.nan                        # np.argmax returns the first NaN
    sig = orc.COCO17_SIGMAS[:K]
    with np.errstate(all="ignore"):
        want_l, want_v, want_c = orc.heatmap_expected_value(hm, sig, "scipy", return_heatmap=True)
    got_l, got_v, got_c = pp.get_heatmap_expected_value(hm, sig, return_heatmap=True)
    np.testing.assert_array_equal(got_c[:7], want_c[:7])
    np.testing.assert_array_equal(got_l[:7], want_l[:7])
    np.testing.assert_array_equal(got_v[:7], want_v[:7])
    assert np.array_equal(got_l[0], [0, 0]) and got_v[0] == 0
    # NaN map: same (NaN-poisoned) peak location as numpy's argmax on the reference map
    assert np.isnan(got_c[7]).any()
    np.testing.assert_array_equal(np.isnan(got_c[7]), np.isnan(want_c[7]))
    np.testing.assert_array_equal(got_l[7], want_l[7])
    # the same maps through the default decode (no convolved map returned) and through the wave-per-map path (its flat
    # and NaN maps take the hand-over list), three times (the list resets itself)
    from probpose_pytorch_amd import _lib, heatmap as hmod
    for flags in (0, _lib.DECODE_WAVE, _lib.DECODE_WAVE, _lib.DECODE_WAVE):
        hmod.DECODE_FLAGS = flags
        try:
            fast_l, fast_v = pp.get_heatmap_expected_value(hm, sig)
        finally:
            hmod.DECODE_FLAGS = 0
        np.testing.assert_array_equal(fast_l, got_l)
        np.testing.assert_array_equal(fast_v, got_v)


def test_empty_batch(pp):
    codec = pp.Codec(pp.ProbMap((192, 256), (48, 64), orc.COCO17_SIGMAS))
    z = torch.zeros((0, 17, 64, 48), device="cuda")
    a = torch.zeros((0, 17, 1, 1), device="cuda")
    (kpts, scores), prob, *_ = codec.decode((z, a, a, a, a))
    assert kpts.shape == (0, 17, 2) and scores.shape == (0, 17) and prob.shape == (0, 1, 17)


def test_packed_record_equals_separate_outputs(pp):
    """The [B,K,7] f64 record the kernel writes for the all-gather == the separately written fields."""
    from probpose_pytorch_amd import parallel
    B, K = 5, 17
    codec = pp.Codec(pp.ProbMap((192, 256), (48, 64), orc.COCO17_SIGMAS))
    hm = torch.from_numpy(orc.synthetic_heatmaps(B, K, 64, 48, seed=3)).cuda()
    g = torch.Generator().manual_seed(9)
    aux = tuple(torch.rand((B, K, 1, 1), generator=g).cuda() for _ in range(4))
    out = codec.decode_device((hm,) + aux)
    assert out["packed"].shape == (B, K, 7) and out["packed"].dtype == torch.float64
    rebuilt = parallel.pack_decoded({k: v for k, v in out.items() if k != "packed"})
    assert torch.equal(out["packed"], rebuilt)
    assert parallel.pack_decoded(out) is out["packed"]


def test_full_size_batch_properties(pp):
    """BASELINE sizes (B=64 K=17 64x48 and 128 crops K=133 96x72): too slow for the
    scipy oracle as a whole, so check size-independent properties + a sampled
    subset against the oracle."""
    for (B, K, H, W, insz) in ((64, 17, 64, 48, (192, 256)), (128, 133, 96, 72, (288, 384))):
        sig = orc.COCO17_SIGMAS if K == 17 else np.random.default_rng(133).uniform(0.02, 0.11, K)
        rng = np.random.default_rng(B)
        base = orc.synthetic_heatmaps(2, K, H, W, seed=B, kind="peaked")
        hm = np.tile(base, (B // 2, 1, 1, 1))                 # crops repeat with period 2
        t = torch.from_numpy(hm).cuda()
        probmap = pp.ProbMap(insz, (W, H), sig)
        out = probmap.decode_device(t)
        kpts, scores = out["kpts"].cpu().numpy(), out["scores"].cpu().numpy()
        # batch independence: identical crops decode identically wherever they sit in the batch
        np.testing.assert_array_equal(kpts[0::2], np.broadcast_to(kpts[0], kpts[0::2].shape))
        np.testing.assert_array_equal(kpts[1::2], np.broadcast_to(kpts[1], kpts[1::2].shape))
        # permutation equivariance over keypoint channels with equal sigma is covered by the
        # oracle comparison on the two distinct crops:
        want_k, want_s = zip(*(orc.probmap_decode(base[b], insz, (W, H), sig) for b in range(2)))
        np.testing.assert_allclose(kpts[:2], np.concatenate(want_k), rtol=0, atol=ATOL_KPTS)
        np.testing.assert_array_equal(scores[:2], np.concatenate(want_s))
        # bounds: sub-pixel shift of an interior strict maximum is within half a pixel ... the
        # decoded point stays inside the rescaled map
        assert np.all(kpts[..., 0] > -0.5 * insz[0] / (W - 1)) and np.all(kpts[..., 0] < insz[0] * (1 + 0.5 / (W - 1)))
        # idempotence of the launch (no hidden state): second run is bit-identical
        again = probmap.decode_device(t)["kpts"].cpu().numpy()
        np.testing.assert_array_equal(again, kpts)


def test_decode_c_entry_without_workspace_takes_the_workgroup_kernels(pp):
    """pp_decode_f32 with workspace = NULL on a 64x48 batch (the integration stub of round 1 passed none): no error, the
    workgroup-per-map kernels run, same numbers as with the hand-over workspace."""
    import ctypes as C
    from probpose_pytorch_amd import _lib
    from probpose_pytorch_amd.heatmap import oks_tap_table
    L = _lib.lib()
    B, K, H, W = 5, 17, 64, 48
    hm = torch.from_numpy(orc.synthetic_heatmaps(B, K, H, W, seed=99, kind="peaked")).cuda()
    hm[1, 2, 10:40, 5:30] = 1.0                                 # a plateau: goes through the hand-over when there is one
    taps, radius = oks_tap_table(K, H, W, orc.COCO17_SIGMAS)
    taps, radius = torch.from_numpy(taps).cuda(), torch.from_numpy(radius).cuda()
    res = []
    for with_ws in (True, False):
        kpts = torch.zeros((B, K, 2), dtype=torch.float64, device="cuda")
        scores = torch.zeros((B, K), device="cuda")
        locs = torch.zeros((B, K, 2), device="cuda")
        ws = torch.zeros((int(L.pp_decode_workspace_bytes(B, K, H, W)),), dtype=torch.uint8, device="cuda")
        rc = L.pp_decode_f32(_lib.ptr(hm), None, None, None, None, B, K, H, W, _lib.ptr(taps), _lib.ptr(radius),
                             float(W - 1), float(H - 1), 192.0, 256.0, _lib.ptr(kpts), _lib.ptr(scores), _lib.ptr(locs),
                             None, None, None, None, _lib.ptr(ws) if with_ws else None, _lib.DECODE_WAVE if with_ws else 0,
                             _lib.stream_ptr())
        _lib.check(rc, "pp_decode_f32")
        res.append((kpts.cpu().numpy(), scores.cpu().numpy(), locs.cpu().numpy()))
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


def test_full_size_wave_decode_equals_all_pixel_decode(pp, monkeypatch):
    """At the sizes the decode rate is quoted on (1024 x 17 x 64x48 and 128 x 133 x 96x72), on DISTINCT device-made maps
    (blobs + noise, some on borders, some zero, some clamped plateaus): the wave-per-map path and the all-pixel
    float64 kernel return identical keypoints, scores and locations for every one of the 17 408 / 17 024 maps."""
    for (B, K, H, W, insz) in ((1024, 17, 64, 48, (192, 256)), (128, 133, 96, 72, (288, 384))):
        g = torch.Generator(device="cuda").manual_seed(B + K)
        yy = torch.arange(H, device="cuda", dtype=torch.float32)[None, None, :, None]
        xx = torch.arange(W, device="cuda", dtype=torch.float32)[None, None, None, :]
        cx = torch.rand((B, K, 1, 1), device="cuda", generator=g) * (W - 1)
        cy = torch.rand((B, K, 1, 1), device="cuda", generator=g) * (H - 1)
        n = torch.arange(B * K, device="cuda").reshape(B, K, 1, 1)
        cx = torch.where(n % 7 == 3, torch.where((n // 7) % 2 == 0, 0.0, W - 1.0), cx)
        sg = 1 + 2 * torch.rand((B, K, 1, 1), device="cuda", generator=g)
        amp = 0.3 + 1.2 * torch.rand((B, K, 1, 1), device="cuda", generator=g)        # > 1: clamped plateaus
        hm = amp * torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
        hm = (hm + 0.02 * torch.rand(hm.shape, device="cuda", generator=g)).clamp_(0, 1)
        hm = torch.where(n % 11 == 5, torch.zeros_like(hm), hm).contiguous()
        sig = orc.COCO17_SIGMAS if K == 17 else np.random.default_rng(133).uniform(0.02, 0.11, K)
        probmap = pp.ProbMap(insz, (W, H), sig)
        from probpose_pytorch_amd import _lib, heatmap as hmod
        monkeypatch.setattr(hmod, "DECODE_FLAGS", 0)
        fast = {k: v.cpu().numpy() for k, v in probmap.decode_device(hm).items()}
        again = {k: v.cpu().numpy() for k, v in probmap.decode_device(hm).items()}     # the work list cleaned itself up
        monkeypatch.setattr(hmod, "DECODE_FLAGS", _lib.DECODE_ALL_PIXEL)
        slow = {k: v.cpu().numpy() for k, v in probmap.decode_device(hm).items()}
        monkeypatch.setattr(hmod, "DECODE_FLAGS", 0)
        for k in ("kpts", "scores", "locs"):
            np.testing.assert_array_equal(again[k], slow[k], err_msg=f"second launch {k} {hm.shape}")
        for k in ("kpts", "scores", "locs"):
            np.testing.assert_array_equal(fast[k], slow[k], err_msg=f"{k} {hm.shape}")
        assert np.isfinite(fast["kpts"]).all()


# ---------------------------------------------------------------------------
# Target generation: ProbMap.encode on the GPU vs goldens minted from the reference
# ---------------------------------------------------------------------------
ENCODE_CASES = [("encode_k17_sigma2", 17, (192, 256), (48, 64), "coco", 2.0, 31),
                ("encode_k17_persigma", 17, (192, 256), (48, 64), "coco", None, 32),
                ("encode_k133_sigma2", 133, (288, 384), (72, 96), "k133", 2.0, 33)]


def _encode_sigmas(tag):
    return orc.COCO17_SIGMAS if tag == "coco" else np.random.default_rng(133).uniform(0.02, 0.11, 133)


@pytest.mark.parametrize("name,K,in_size,hm_size,sig,sigma,seed", ENCODE_CASES)
def test_encode_matches_reference_golden(pp, name, K, in_size, hm_size, sig, sigma, seed):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    kp, vis = orc.synthetic_keypoints(K, in_size, seed)
    enc = pp.ProbMap(in_size, hm_size, _encode_sigmas(sig), sigma=sigma).encode(kp.copy(), vis.copy())
    n = g["heatmaps"].shape[0]
    assert enc["heatmaps"].dtype == np.float32 and enc["heatmaps"].shape == (K, hm_size[1], hm_size[0])
    assert np.array_equal(enc["heatmaps"][:n], g["heatmaps"])          # bit-exact float32 maps
    assert np.array_equal(enc["keypoint_weights"], g["weights"])
    assert np.array_equal(enc["in_image"], g["in_image"]) and np.array_equal(enc["annotated"], g["annotated"])
    assert np.array_equal(enc["heatmap_keypoints"], g["heatmap_keypoints"])


def test_encode_batched_equals_per_crop_oracle(pp):
    pm = pp.ProbMap((192, 256), (48, 64), orc.COCO17_SIGMAS, sigma=None)
    kps, viss = zip(*[orc.synthetic_keypoints(17, (192, 256), 100 + i) for i in range(5)])
    heat, wts = pm.encode_device(np.concatenate(kps), np.concatenate(viss))
    heat, wts = heat.cpu().numpy(), wts.cpu().numpy()
    for i in range(5):
        ref = orc.probmap_encode(kps[i], viss[i], (192, 256), (48, 64), orc.COCO17_SIGMAS, None)
        assert np.array_equal(heat[i], ref["heatmaps"]) and np.array_equal(wts[i:i + 1], ref["keypoint_weights"])
    assert pm.encode_device(np.zeros((0, 17, 2), np.float32))[0].shape == (0, 17, 64, 48)


def test_screened_decode_equals_all_pixel_float64_decode(pp, monkeypatch):
    """The default decode evaluates the float64 convolution only at the float32-screened candidates: one wave per map
    (decode_wave_kernel) on 64x48 and 96x72 maps, one workgroup per map (decode_screen_kernel; flags PP_DECODE_NO_WAVE |
    PP_DECODE_SCREEN force it everywhere) otherwise; PP_DECODE_ALL_PIXEL selects the all-pixel float64 kernel.
    All three must return identical numbers on random, peaked, flat, saturated, tied, negative, non-finite and tiny
    maps."""
    import torch
    rng = np.random.default_rng(77)
    cases = []
    cases.append(orc.synthetic_heatmaps(6, 17, 64, 48, seed=5, kind="peaked"))
    cases.append(orc.synthetic_heatmaps(3, 17, 64, 48, seed=6, kind="uniform"))
    cases.append(orc.synthetic_heatmaps(2, 133, 96, 72, seed=7, kind="peaked"))
    cases.append(orc.synthetic_heatmaps(1, 133, 96, 72, seed=8, kind="uniform"))

    def hard(H, W):
        sat = np.zeros((2, 17, H, W), np.float32)
        sat[:, :, 10:50, 5:40] = 1.0                                    # clamped plateau: hundreds of exact ties
        sat[1, 3] = 0.25                                                 # constant map
        sat[1, 4] = 0.0
        sat[1, 5, 0, 0] = 1.0                                            # corner peak
        sat[1, 6, H - 1, W - 1] = 1.0
        sat[1, 7] = rng.random((H, W), dtype=np.float32) * 1e-6          # tiny values: threshold scales with max |x|
        sat[1, 8] = -rng.random((H, W), dtype=np.float32)                # negative map
        sat[1, 9, 20, 20] = np.nan
        sat[1, 10, 5, 5] = np.inf
        sat[1, 11] = np.round(rng.random((H, W), dtype=np.float32) * 3) / 3   # many exact ties at several levels
        sat[1, 12, H - 1, 7] = sat[1, 12, 0, W - 2] = 0.5                # equal peaks on two borders
        sat[1, 13, H // 2, W // 2] = sat[1, 13, H // 2, W // 2 + 1] = 0.75   # two-pixel ridge: > 1 candidate
        sat[1, 14, 63 % H, 3] = sat[1, 14, (64 % H), 3] = 1.0            # straddles the 64-row round of the big maps
        sat[1, 15, 30, 35 % W] = sat[1, 15, 30, 36 % W] = 1.0            # straddles the 36-column round
        return sat

    cases.append(hard(64, 48))
    cases.append(hard(96, 72))
    cases.append(rng.random((2, 5, 7, 5), dtype=np.float32))            # smaller than the kernel radius
    cases.append(rng.random((1, 3, 33, 27), dtype=np.float32))          # W % 4 != 0
    for hm in cases:
        B, K, H, W = hm.shape
        sig = (orc.COCO17_SIGMAS if K == 17 else np.random.default_rng(K).uniform(0.02, 0.11, K))
        codec = pp.Codec(pp.ProbMap((4 * W, 4 * H), (W, H), sig))
        aux = [rng.random((B, K, 1, 1), dtype=np.float32) for _ in range(4)]
        pred = tuple(torch.from_numpy(a).cuda() for a in (hm, *aux))
        from probpose_pytorch_amd import _lib, heatmap as hmod
        monkeypatch.setattr(hmod, "DECODE_FLAGS", 0)
        legs = {"default": codec.decode(pred)}
        monkeypatch.setattr(hmod, "DECODE_FLAGS", _lib.DECODE_WAVE)       # the wave-per-map kernel also at small batch sizes
        legs["wave"] = codec.decode(pred)
        legs["wave again"] = codec.decode(pred)                            # (its hand-over list reset itself)
        monkeypatch.setattr(hmod, "DECODE_FLAGS", _lib.DECODE_NO_WAVE | _lib.DECODE_SCREEN)   # workgroup-per-map screened form
        legs["screened"] = codec.decode(pred)
        monkeypatch.setattr(hmod, "DECODE_FLAGS", _lib.DECODE_ALL_PIXEL)
        slow = codec.decode(pred)
        monkeypatch.setattr(hmod, "DECODE_FLAGS", 0)
        for name, fast in legs.items():
            np.testing.assert_array_equal(fast[0][0], slow[0][0], err_msg=f"{name} {hm.shape}")
            np.testing.assert_array_equal(fast[0][1], slow[0][1], err_msg=f"{name} {hm.shape}")
            for a, b in zip(fast[1:], slow[1:]):
                np.testing.assert_array_equal(a, b)
