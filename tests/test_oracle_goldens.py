"""CPU tests: the oracle restatement against the goldens minted from the
reference (tests/golden/make_goldens.py), plus host-side logic."""
import os

import numpy as np
import pytest

from oracle import probpose_oracle as orc
from tests.helpers import DECODE_FIXTURES, GOLDEN, load_decode_fixture, sha


@pytest.mark.parametrize("name", DECODE_FIXTURES)
@pytest.mark.parametrize("backend", ["scipy", "torch"])
def test_oracle_decode_matches_reference(name, backend):
    g, hm, aux = load_decode_fixture(name)
    (kpts, scores), prob, vis, oks, err = orc.codec_decode(
        (hm, *aux), tuple(g["in_size"]), (int(g["W"]), int(g["H"])), g["sigmas"], backend=backend)
    assert kpts.dtype == np.float64 and scores.dtype == np.float32 and err.dtype == np.float64
    np.testing.assert_array_equal(kpts, g["kpts"])      # bit-exact: same arithmetic, same libs
    np.testing.assert_array_equal(scores, g["scores"])
    np.testing.assert_array_equal(prob, g["prob"])
    np.testing.assert_array_equal(vis, g["vis"])
    np.testing.assert_array_equal(oks, g["oks"])
    np.testing.assert_array_equal(err, g["err"])


def test_oracle_convmaps_match_reference_both_backends():
    g = np.load(os.path.join(GOLDEN, "convmaps_k17.npz"))
    hm = orc.synthetic_heatmaps(1, 17, 64, 48, 4321, "peaked")[0]
    assert sha(hm) == str(g["hm_sha"])
    for backend, key in (("scipy", "conv_scipy"), ("torch", "conv_torch")):
        locs, vals, conv = orc.heatmap_expected_value(hm, orc.COCO17_SIGMAS, backend, return_heatmap=True)
        np.testing.assert_array_equal(conv, g[key])
        np.testing.assert_array_equal(locs, g["locs"])
        np.testing.assert_array_equal(vals, g["vals"])
    # the reference's own test asserts rtol=1e-5 between its back-ends (tests/test_heatmap.py:12)
    np.testing.assert_allclose(g["conv_scipy"], g["conv_torch"], rtol=1e-5, atol=1e-8)


def test_oracle_bigmap_reference_test_shape():
    g = np.load(os.path.join(GOLDEN, "bigmap_256.npz"))
    rng = np.random.default_rng(2024)
    big = rng.random((20, 256, 256), dtype=np.float32)
    sig = rng.random(20, dtype=np.float32)
    assert sha(big) == str(g["hm_sha"])
    np.testing.assert_array_equal(sig, g["sigmas"])
    locs, vals, conv = orc.heatmap_expected_value(big, sig, "scipy", return_heatmap=True)
    np.testing.assert_array_equal(locs, g["locs"])
    np.testing.assert_array_equal(vals, g["vals"])
    np.testing.assert_array_equal(conv[:, ::16, ::16], g["conv_sample"])


def test_subpixel_border_and_flat_cases():
    conv = np.zeros((3, 8, 8), np.float32)
    conv[0, 0, 3] = 1.0            # border peak: left at the integer location (heatmap.py:120-125)
    conv[1, 4, 4] = 1.0            # isolated interior peak: symmetric -> zero shift
    conv[2, 4, 4], conv[2, 4, 5] = 1.0, 0.5
    locs = np.array([[3, 0], [4, 4], [4, 4]], np.float32)
    out = orc.subpixel_refine(conv, locs)
    np.testing.assert_array_equal(out[0], [3, 0])
    np.testing.assert_array_equal(out[1], [4, 4])
    assert out[2, 0] == np.float32(4) + np.float32(-(0.25) / (-1.5)) and out[2, 1] == 4


def test_host_tap_table_is_the_separable_factor_of_the_reference_kernel():
    """The product's 1-D taps (probpose_pytorch_amd.heatmap.oks_tap_table) must
    reproduce the reference's 2-D kernels (heatmap.py:170-194) as an outer product."""
    from probpose_pytorch_amd.heatmap import oks_tap_table
    for sig in (orc.COCO17_SIGMAS, orc.COCO17_SIGMAS.astype(np.float32), np.array([0.5, 0.05, 0.001, 0.2]),
                np.random.default_rng(0).random(20, dtype=np.float32)):
        for (H, W) in ((64, 48), (96, 72), (256, 256), (7, 5)):
            K = len(sig)
            taps, radius = oks_tap_table(K, H, W, sig)
            kerns = orc.oks_kernels(K, H, W, sig)
            for k in range(K):
                r = radius[k]
                assert kerns[k].shape == (1, 2 * r + 1, 2 * r + 1)
                assert 2 <= r <= 9
                t = taps[k, :2 * r + 1]
                assert np.all(taps[k, 2 * r + 1:] == 0)
                np.testing.assert_allclose(np.outer(t, t), kerns[k][0], rtol=1e-13, atol=0)
                assert abs(t.sum() - 1) < 1e-15


def test_separable_f64_conv_rounds_to_the_same_f32_map():
    """Design check for the HIP kernel: row pass + column pass in float64 with
    the 1-D taps rounds to the same float32 convolved map as scipy's 2-D
    float64 accumulation (differences are ~1e-16 relative, far below f32 ulp)."""
    from probpose_pytorch_amd.heatmap import oks_tap_table
    g = np.load(os.path.join(GOLDEN, "convmaps_k17.npz"))
    hm = orc.synthetic_heatmaps(1, 17, 64, 48, 4321, "peaked")[0]
    taps, radius = oks_tap_table(17, 64, 48, orc.COCO17_SIGMAS)
    H, W = 64, 48
    for k in range(17):
        r = radius[k]
        t = taps[k, :2 * r + 1]
        src = hm[k].astype(np.float64)
        pad = np.pad(src, ((0, 0), (r, r)), mode="symmetric")
        rows = sum(t[j] * pad[:, j:j + W] for j in range(2 * r + 1))
        pad = np.pad(rows, ((r, r), (0, 0)), mode="symmetric")
        conv = sum(t[j] * pad[j:j + H, :] for j in range(2 * r + 1)).astype(np.float32)
        np.testing.assert_array_equal(conv, g["conv_scipy"][k])


def test_vit_restatement_self_consistency():
    """Backbone parity is unpinned by the reference (timm absent): check the
    restatement against an independent formulation (explicit softmax vs SDPA)
    and against float64."""
    import torch
    from probpose_pytorch_amd.synthetic import synthetic_vit_state
    sd = synthetic_vit_state(img_size=(64, 48), patch=16, embed_dim=96, depth=2, seed=3)
    x = torch.rand(2, 3, 64, 48, generator=torch.Generator().manual_seed(5))
    a = orc.vit_forward_features(sd, x, patch=16, heads=4)
    b = orc.vit_forward_features(sd, x, patch=16, heads=4, explicit_softmax=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    c = orc.vit_forward_features(sd64, x.double(), patch=16, heads=4, explicit_softmax=True)
    assert a.shape == (2, 12, 96)
    torch.testing.assert_close(a, b, rtol=0, atol=2e-5)
    torch.testing.assert_close(a.double(), c, rtol=0, atol=2e-5)


def test_head_restatement_matches_reference_golden():
    import torch
    from probpose_pytorch_amd.synthetic import synthetic_features, synthetic_head_state
    g = np.load(os.path.join(GOLDEN, "head_c384.npz"))
    C, K = int(g["C"]), int(g["K"])
    pools = [tuple(int(v) for v in p) for p in g["pools"]]
    sd = synthetic_head_state(C, K, n_pools=len(pools), deconv_out=(256, 256), seed=11)
    np.testing.assert_array_equal(sd["final_layer.weight"].numpy().ravel()[:8], g["w_probe"])
    feats = synthetic_features(2, C, 16, 12, seed=12)
    assert sha(feats.numpy()) == str(g["feats_sha"])
    with torch.no_grad():
        out = orc.head_forward(sd, feats, pools=pools)
    for o, key in zip(out, ("heatmaps", "prob", "vis", "oks", "err")):
        np.testing.assert_allclose(o.numpy(), g[key], rtol=0, atol=1e-6)
    # chained decode on the *golden* heatmaps reproduces the reference's decoded keypoints
    (kpts, scores), *_rest, err = orc.codec_decode(
        tuple(g[k] for k in ("heatmaps", "prob", "vis", "oks", "err")), (192, 256), (48, 64),
        orc.COCO17_SIGMAS)
    np.testing.assert_array_equal(kpts, g["kpts"])
    np.testing.assert_array_equal(scores, g["scores"])
    np.testing.assert_array_equal(err, g["dec_err"])


@pytest.mark.parametrize("name,K,in_size,hm_size,sig,sigma,seed", [
    ("encode_k17_sigma2", 17, (192, 256), (48, 64), "coco", 2.0, 31),
    ("encode_k17_persigma", 17, (192, 256), (48, 64), "coco", None, 32),
    ("encode_k133_sigma2", 133, (288, 384), (72, 96), "k133", 2.0, 33)])
def test_encode_restatement_equals_reference_golden(name, K, in_size, hm_size, sig, sigma, seed):
    """G6: oracle.probmap_encode vs the imported reference's ProbMap.encode (codec.py:138-212)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sigmas = orc.COCO17_SIGMAS if sig == "coco" else np.random.default_rng(133).uniform(0.02, 0.11, 133)
    kp, vis = orc.synthetic_keypoints(K, in_size, seed)
    e = orc.probmap_encode(kp, vis, in_size, hm_size, sigmas, sigma)
    n = g["heatmaps"].shape[0]
    assert np.array_equal(e["heatmaps"][:n], g["heatmaps"])
    assert np.array_equal(e["keypoint_weights"], g["weights"]) and np.array_equal(e["in_image"], g["in_image"])
    assert np.array_equal(e["heatmap_keypoints"], g["heatmap_keypoints"])


HEAD_VARIANTS = {
    # name: (C, K, pools, feature hw, deconv_out, conv_out, conv_kernels, final_kernel, seed) -- as minted by
    # tests/golden/make_goldens_head_variants.py from the imported reference head
    "A": (128, 17, [(4, 3), (2, 2), (2, 2)], (16, 12), (64, 64), (64,), (3,), 3, 21),
    "B": (128, 17, [(4, 3), (2, 2), (2, 2)], (16, 12), (64,), (64, 17), (3, 1), None, 22),
    "C": (64, 5, [(4, 3), (2, 2)], (8, 6), (), (64,), (1,), 1, 23),
}


@pytest.mark.parametrize("name", sorted(HEAD_VARIANTS))
def test_head_restatement_variants_match_reference_golden(name):
    """The constructor branches beside the default head: conv stack (head.py:407-431), final kernel 3, Identity final
    layer (head.py:234-235), no deconvolution -- oracle vs the reference ProbMapHead's own outputs."""
    import torch
    from probpose_pytorch_amd.synthetic import synthetic_features, synthetic_head_state
    C, K, pools, (h, w), dec, conv, ck, fk, seed = HEAD_VARIANTS[name]
    g = np.load(os.path.join(GOLDEN, "head_variants.npz"))
    sd = synthetic_head_state(C, K, n_pools=len(pools), deconv_out=dec, seed=seed, final_kernel=fk, conv_out=conv,
                              conv_kernels=ck)
    feats = synthetic_features(2, C, h, w, seed=seed + 100)
    with torch.no_grad():
        out = orc.head_forward(sd, feats, pools=pools, n_deconv=len(dec), final_kernel=fk, conv_kernels=ck)
    for o, key in zip(out, ("heatmaps", "prob", "vis", "oks", "err")):
        np.testing.assert_allclose(o.numpy(), g[f"{name}_{key}"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("C,depth,heads,img", [(96, 2, 4, (64, 48)), (384, 3, 12, (64, 48)), (128, 2, 2, (96, 64))])
def test_vit_restatement_against_torch_transformer_encoder(C, depth, heads, img):
    """Backbone parity cannot be pinned on the reference (its ViT is timm's, not importable, and the reference holds no
    fixture for it).  The next best thing is an INDEPENDENT implementation of the same architecture from a library that
    is present: torch.nn.TransformerEncoderLayer(norm_first=True, activation='gelu', layer_norm_eps=1e-6) is the same
    pre-LN block (x += MHA(LN1 x); x += fc2(GELU(fc1(LN2 x)))) with the same [3][heads][head_dim] in-projection layout
    as timm's fused qkv Linear (backbone.py:26-33 builds it with timm defaults: qkv_bias, exact-erf GELU, no
    LayerScale).  Weights copied from the timm-named state_dict; patch embedding = Conv2d(3, C, 16, stride 16), learned
    position table added, final LayerNorm: outputs agree with oracle.vit_forward_features to 3e-5."""
    import torch
    from torch import nn
    from probpose_pytorch_amd.synthetic import synthetic_vit_state
    sd = synthetic_vit_state(img_size=img, patch=16, embed_dim=C, depth=depth, seed=7)
    x = torch.rand(2, 3, *img, generator=torch.Generator().manual_seed(8))
    layers = []
    for i in range(depth):
        lyr = nn.TransformerEncoderLayer(d_model=C, nhead=heads, dim_feedforward=4 * C, dropout=0.0, activation="gelu",
                                         layer_norm_eps=1e-6, batch_first=True, norm_first=True)
        p = f"blocks.{i}."
        with torch.no_grad():
            lyr.self_attn.in_proj_weight.copy_(sd[p + "attn.qkv.weight"])
            lyr.self_attn.in_proj_bias.copy_(sd[p + "attn.qkv.bias"])
            lyr.self_attn.out_proj.weight.copy_(sd[p + "attn.proj.weight"])
            lyr.self_attn.out_proj.bias.copy_(sd[p + "attn.proj.bias"])
            lyr.norm1.weight.copy_(sd[p + "norm1.weight"]); lyr.norm1.bias.copy_(sd[p + "norm1.bias"])
            lyr.norm2.weight.copy_(sd[p + "norm2.weight"]); lyr.norm2.bias.copy_(sd[p + "norm2.bias"])
            lyr.linear1.weight.copy_(sd[p + "mlp.fc1.weight"]); lyr.linear1.bias.copy_(sd[p + "mlp.fc1.bias"])
            lyr.linear2.weight.copy_(sd[p + "mlp.fc2.weight"]); lyr.linear2.bias.copy_(sd[p + "mlp.fc2.bias"])
        layers.append(lyr)
    pe = nn.Conv2d(3, C, 16, stride=16)
    norm = nn.LayerNorm(C, eps=1e-6)
    with torch.no_grad():
        pe.weight.copy_(sd["patch_embed.proj.weight"]); pe.bias.copy_(sd["patch_embed.proj.bias"])
        norm.weight.copy_(sd["norm.weight"]); norm.bias.copy_(sd["norm.bias"])
    for train_mode in (False, True):           # eval takes torch's fused fast path, train mode (dropout 0) the plain one
        with torch.no_grad():
            t = pe(x).flatten(2).transpose(1, 2) + sd["pos_embed"]
            for lyr in layers:
                lyr.train(train_mode)
                t = lyr(t)
            want = norm(t)
            got = orc.vit_forward_features(sd, x, patch=16, heads=heads)
        assert got.shape == want.shape == (2, (img[0] // 16) * (img[1] // 16), C)
        torch.testing.assert_close(got, want, rtol=0, atol=3e-5)
