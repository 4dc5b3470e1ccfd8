"""End-to-end GPU parity of the forward path (backbone -> head -> decode)
against the CPU oracle / the goldens minted from the reference.

Staged parity claim ("bf16 vs 1e-4"):
  (i)   decode kernel vs CPU decode on identical heatmaps: exact (test_decode_gpu.py)
  (ii)  fp32 GPU forward vs CPU forward: <= 1e-4 abs on heatmaps / aux outputs,
        and decoded keypoints <= 1e-4 px wherever the argmax is well separated
  (iii) bf16: measured deviation is reported and loosely bounded.
"""
import os

import numpy as np
import pytest
import torch

from oracle import probpose_oracle as orc
from tests.helpers import GOLDEN, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg(built_lib):
    assert torch.cuda.is_available()
    import probpose_pytorch_amd as p
    from probpose_pytorch_amd import backbone, head, model, synthetic
    return dict(p=p, backbone=backbone, head=head, model=model, syn=synthetic)


def _build(pkg, img, C, depth, heads, K, pools, deconv=(256, 256), seed=0):
    syn = pkg["syn"]
    m = pkg["model"].ProbPoseModel(
        pkg["backbone"].ScratchViTBackbone(img, 16, embed_dim=C, depth=depth, num_heads=heads),
        pkg["head"].ProbMapHead(C, K, pools, deconv, (4,) * len(deconv), final_layer_kernel_size=1))
    sd = syn.synthetic_model_state(img, 16, C, depth, K, len(pools), deconv, seed=seed)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


# ---- observed end-to-end agreement, recorded and asserted against the committed observation --------------------
# The share of decoded keypoints that agree with the CPU path to 1e-4 px (the rest are verified near-ties of the
# random-weight heatmaps, see _assert_keypoints_match_up_to_near_ties) is written to gpurun_out/ on the GPU box and,
# copied to profiles/r02_parity.json, becomes the threshold of later runs: observed - 0.05 (0.7 for a case that
# has no committed observation yet).
_PARITY_COMMITTED = os.path.join(os.path.dirname(GOLDEN), "..", "profiles", "r02_parity.json")
_PARITY_OBSERVED = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out", "r02_parity_observed.json")


def _check_and_record_fraction(name: str, frac: float, extra=None):
    import json
    committed = {}
    if os.path.exists(_PARITY_COMMITTED):
        with open(_PARITY_COMMITTED) as f:
            committed = json.load(f).get("near_tie_free_fraction", {})
    floor = committed[name] - 0.05 if name in committed else 0.7
    try:
        os.makedirs(os.path.dirname(_PARITY_OBSERVED), exist_ok=True)
        obs = {}
        if os.path.exists(_PARITY_OBSERVED):
            with open(_PARITY_OBSERVED) as f:
                obs = json.load(f)
        obs.setdefault("near_tie_free_fraction", {})[name] = round(float(frac), 4)
        if extra:
            obs.setdefault("detail", {})[name] = extra
        with open(_PARITY_OBSERVED, "w") as f:
            json.dump(obs, f, indent=1, sort_keys=True)
    except OSError:
        pass
    assert frac >= floor, f"{name}: {frac:.4f} of keypoints within 1e-4 px, committed floor {floor:.4f}"


def _assert_keypoints_match_up_to_near_ties(pkg, gpu_heat, ref_heat, gpu_kpts, ref_kpts, sigmas, tol=1e-4):
    """Heatmaps agree to `tol`, so the convolved maps do too (the OKS kernels are normalised): a GPU
    peak may differ from the CPU peak only where the reference map holds a near-tie, i.e. the GPU
    arg-max must be a (2*tol + ulp)-arg-max of the reference convolved map.  Everywhere else the
    decoded keypoints must agree to 1e-4 px."""
    d = np.abs(gpu_kpts - ref_kpts).max(-1)
    bad = np.argwhere(d > 1e-4)
    frac = 1.0 - len(bad) / d.size
    for b, k in bad:
        _, _, conv_ref = orc.heatmap_expected_value(ref_heat[b], sigmas, "scipy", return_heatmap=True)
        _, _, conv_gpu = pkg["p"].get_heatmap_expected_value(gpu_heat[b], sigmas, return_heatmap=True)
        peak_gpu = np.unravel_index(np.argmax(conv_gpu[k]), conv_gpu[k].shape)
        gap = conv_ref[k].max() - conv_ref[k][peak_gpu]
        assert gap <= 2.5 * tol, f"crop {b} keypoint {k}: GPU peak is {gap:.3g} below the reference maximum"
    return frac


def test_head_fp32_matches_reference_golden(pkg):
    """G3: HIP ProbMapHead vs the reference ProbMapHead outputs (fp32, atol 1e-4)."""
    g = np.load(os.path.join(GOLDEN, "head_c384.npz"))
    C, K = int(g["C"]), int(g["K"])
    pools = [tuple(int(v) for v in p) for p in g["pools"]]
    head = pkg["head"].ProbMapHead(C, K, pools, (256, 256), (4, 4), final_layer_kernel_size=1)
    head.load_state_dict(pkg["syn"].synthetic_head_state(C, K, len(pools), (256, 256), seed=11))
    head = head.cuda().eval()
    feats = pkg["syn"].synthetic_features(2, C, 16, 12, seed=12)
    assert sha(feats.numpy()) == str(g["feats_sha"])
    out = head(feats.cuda())
    assert out[0].shape == (2, K, 64, 48) and out[1].shape == (2, K, 1, 1) and out[0].dtype == torch.float32
    for o, key in zip(out, ("heatmaps", "prob", "vis", "oks", "err")):
        np.testing.assert_allclose(o.cpu().numpy(), g[key], rtol=0, atol=1e-4)
    # G4: chained decode; the HIP decode of the HIP heatmaps equals the oracle decode of the same heatmaps
    codec = pkg["p"].Codec(pkg["p"].ProbMap((192, 256), (48, 64), orc.COCO17_SIGMAS))
    got = codec.decode(out)
    want = orc.codec_decode([o.cpu().numpy() for o in out], (192, 256), (48, 64), orc.COCO17_SIGMAS)
    np.testing.assert_array_equal(got[0][0], want[0][0])
    np.testing.assert_array_equal(got[0][1], want[0][1])
    # ... and agrees with the reference's decoded keypoints wherever fp32 reordering did not flip a near-tie
    frac = _assert_keypoints_match_up_to_near_ties(pkg, out[0].cpu().numpy(), g["heatmaps"], got[0][0], g["kpts"],
                                                   orc.COCO17_SIGMAS)
    print(f"\nhead golden: {frac:.2%} of keypoints within 1e-4 px of the reference, the rest are near-ties")
    _check_and_record_fraction("head_golden_c384", frac)


@pytest.mark.parametrize("name", ["A", "B", "C"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_head_variants_match_reference_golden(pkg, name, dtype):
    """ProbMapHead branches beside the default: a conv stack (head.py:407-431), final_layer_kernel_size 3 and None
    (nn.Identity, head.py:234-235), no deconvolution -- the HIP head vs the REFERENCE head's outputs
    (tests/golden/head_variants.npz, minted by make_goldens_head_variants.py): fp32 mode <= 1e-4, bf16 bounded."""
    from tests.test_oracle_goldens import HEAD_VARIANTS
    C, K, pools, (h, w), dec, conv, ck, fk, seed = HEAD_VARIANTS[name]
    g = np.load(os.path.join(GOLDEN, "head_variants.npz"))
    head = pkg["head"].ProbMapHead(C, K, pools, dec, (4,) * len(dec), conv_out_channels=conv or None,
                                   conv_kernel_sizes=ck or None, final_layer_kernel_size=fk)
    head.load_state_dict(pkg["syn"].synthetic_head_state(C, K, n_pools=len(pools), deconv_out=dec, seed=seed, final_kernel=fk,
                                                         conv_out=conv, conv_kernels=ck))
    head = head.cuda().eval().set_compute_dtype(dtype)
    feats = pkg["syn"].synthetic_features(2, C, h, w, seed=seed + 100)
    out = head(feats.cuda())
    up = 2 ** len(dec)
    assert out[0].shape == (2, K, h * up, w * up) and out[0].dtype == torch.float32
    for o, key in zip(out, ("heatmaps", "prob", "vis", "oks", "err")):
        d = np.abs(o.cpu().numpy() - g[f"{name}_{key}"])
        if dtype == torch.float32:
            assert d.max() <= 1e-4, (name, key, d.max())
        else:
            assert d.mean() < 0.02 and d.max() < 0.5, (name, key, d.mean(), d.max())


@pytest.mark.parametrize("cfg", [
    dict(img=(64, 48), C=128, depth=2, heads=2, K=17, pools=[(4, 3)], deconv=(64, 64), B=3),
    dict(img=(256, 192), C=384, depth=12, heads=12, K=17, pools=[(4, 3), (2, 2), (2, 2)], deconv=(256, 256), B=1),
    dict(img=(256, 192), C=768, depth=12, heads=12, K=17, pools=[(4, 3), (2, 2), (2, 2)], deconv=(256, 256), B=2),
    # BASELINE.json configs 3 / 5 dimensions (ViT-L: C 1024, depth 24, 16 heads) and config 4 (ViT-H 384x288, K = 133,
    # N = 432 tokens -> streaming attention with head_dim 80, 96x72 heatmaps, pools (4,3),(2,2),(3,3))
    dict(img=(256, 192), C=1024, depth=24, heads=16, K=17, pools=[(4, 3), (2, 2), (2, 2)], deconv=(256, 256), B=2),
    dict(img=(384, 288), C=1280, depth=32, heads=16, K=133, pools=[(4, 3), (2, 2), (3, 3)], deconv=(256, 256), B=1),
], ids=["tiny", "S1-vit-s-256x192", "vit-b-256x192", "vit-l-256x192", "vit-h-384x288-k133"])
def test_model_fp32_matches_cpu_oracle(pkg, cfg):
    m, sd = _build(pkg, cfg["img"], cfg["C"], cfg["depth"], cfg["heads"], cfg["K"], cfg["pools"], cfg["deconv"])
    x = pkg["syn"].synthetic_crops(cfg["B"], *cfg["img"], seed=1234)
    with torch.no_grad():
        want = orc.model_forward(sd, x, patch=16, heads=cfg["heads"], pools=cfg["pools"], n_deconv=len(cfg["deconv"]))
        got = m(x.cuda())
    # backbone alone, NCHW contract of backbone.py:35-40
    feats = m.backbone(x.cuda())
    want_f = orc.backbone_forward(sd, x, patch=16, heads=cfg["heads"], prefix="backbone.model.")
    assert feats.shape == want_f.shape and feats.is_contiguous()
    np.testing.assert_allclose(feats.cpu().numpy(), want_f.numpy(), rtol=0, atol=1e-4)
    for gt, wt, name in zip(got, want, ("heatmaps", "prob", "vis", "oks", "err")):
        assert gt.shape == wt.shape, name
        np.testing.assert_allclose(gt.cpu().numpy(), wt.numpy(), rtol=0, atol=1e-4, err_msg=name)
    H, W = cfg["img"]
    import bench
    sig = bench.sigmas_for(cfg["K"])
    codec = pkg["p"].Codec(pkg["p"].ProbMap((W, H), (W // 4, H // 4), sig))
    dec = codec.decode(got)
    ref = orc.codec_decode([w.numpy() for w in want], (W, H), (W // 4, H // 4), sig)
    for a, b in zip(dec[1:4], ref[1:4]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    np.testing.assert_allclose(dec[4], ref[4], rtol=0, atol=1e-4)
    frac = _assert_keypoints_match_up_to_near_ties(pkg, got[0].cpu().numpy(), want[0].numpy(), dec[0][0], ref[0][0], sig)
    print(f"\n[{cfg['C']}] keypoints within 1e-4 px of the CPU path: {frac:.2%} (the rest are verified near-ties)")
    d = np.abs(dec[0][0] - ref[0][0]).max(-1)
    _check_and_record_fraction(f"fp32_C{cfg['C']}_{H}x{W}_K{cfg['K']}", frac,
                               dict(heatmap_abs_max=float((got[0].cpu() - want[0]).abs().max()),
                                    kpt_px_median=float(np.median(d)), kpt_px_max=float(d.max()),
                                    keypoints=int(d.size)))


def test_model_bf16_deviation_is_bounded_and_reported(pkg):
    cfg = dict(img=(256, 192), C=384, depth=12, heads=6, K=17, pools=[(4, 3), (2, 2), (2, 2)], deconv=(256, 256))
    m, sd = _build(pkg, cfg["img"], cfg["C"], cfg["depth"], cfg["heads"], cfg["K"], cfg["pools"], cfg["deconv"])
    x = pkg["syn"].synthetic_crops(2, *cfg["img"], seed=1234)
    with torch.no_grad():
        want = orc.model_forward(sd, x, patch=16, heads=cfg["heads"], pools=cfg["pools"])
        m.set_compute_dtype(torch.bfloat16)
        got = m(x.cuda())
    dh = (got[0].cpu() - want[0]).abs()
    da = max((g.cpu() - w).abs().max().item() for g, w in zip(got[1:], want[1:]))
    print(f"\nbf16 vs fp32 CPU: heatmap max|d| {dh.max():.4f} mean|d| {dh.mean():.5f}; aux max|d| {da:.4f}")
    assert dh.mean() < 0.02 and da < 0.1


def test_model_rebuilds_plan_when_weights_change(pkg):
    m, sd = _build(pkg, (64, 48), 128, 1, 2, 5, [(4, 3)], (64,))
    x = pkg["syn"].synthetic_crops(1, 64, 48).cuda()
    a = m(x)[0].clone()
    with torch.no_grad():
        m.head.final_layer.bias.add_(0.25)
    b = m(x)[0]
    assert not torch.equal(a, b)
    with pytest.raises(RuntimeError):
        m.head.train()
        m(x)


def test_inference_cli_end_to_end(pkg, tmp_path, capsys):
    """The reference CLI's call sequence (inference.py:61-112) on a synthetic crop, fp32 parity mode,
    against the oracle with the same weights."""
    from probpose_pytorch_amd import inference
    preds = inference.main(["--backbone", "vit_s", "--input_size", "192,256", "--output", str(tmp_path)])
    assert (tmp_path / "heatmap_16.npy").exists()
    sd = pkg["syn"].synthetic_model_state((256, 192), 16, 384, 12, 17, 3, (256, 256), seed=0)
    x = pkg["syn"].synthetic_crops(1, 256, 192, seed=1234)
    with torch.no_grad():
        want = orc.model_forward(sd, x, patch=16, heads=12, pools=inference.default_pools((16, 12)))
    ref = orc.codec_decode([w.numpy() for w in want], (192, 256), (48, 64), np.array([0.05] * 17))
    assert preds[0][0].shape == (1, 17, 2)
    for a, b in zip(preds[1:], ref[1:]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    assert "Predictions:" in capsys.readouterr().out


def test_full_size_forward_properties_vit_b_bs64(pkg):
    """BASELINE.json configs[1] (ViT-B 256x192 K=17 bf16, batch 64) is too slow for the CPU oracle as a
    whole; check size-independent properties of the full-size GPU forward instead:
    batch independence (a crop's outputs do not depend on its batch neighbours or position), determinism
    (no atomics / races: replays are bit-identical), HIP-graph replay == eager, and a 2-crop subset against
    the CPU oracle within the bf16 deviation bound."""
    import bench
    cfg = dict(bench.CONFIGS["vit_b"])
    model, codec, sd = bench.build(cfg, torch.bfloat16, torch.device("cuda", 0))
    x = pkg["syn"].synthetic_crops(64, 256, 192, seed=77).cuda()
    with torch.no_grad():
        full = [o.clone() for o in model(x)]
        again = model(x)
        for a, b in zip(full, again):
            assert torch.equal(a, b)                                   # deterministic
        perm = torch.arange(63, -1, -1, device="cuda")
        rev = model(x[perm].contiguous())
        for a, b in zip(full, rev):
            assert torch.equal(a[perm], b)                             # position / neighbour independent
        sub = model(x[10:12].contiguous())
        for a, b in zip(full, sub):
            assert torch.equal(a[10:12], b)                            # batch-size independent
        # graph replay == eager
        static_x = x.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            model(static_x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out_g = model(static_x)
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(full, out_g):
            assert torch.equal(a, b)
        want = orc.model_forward(sd, x[:2].cpu(), patch=16, heads=12, pools=cfg["pools"])
    dh = (full[0][:2].cpu() - want[0]).abs()
    assert dh.mean() < 0.03, dh.mean()
    dec = codec.decode(tuple(f[:2] for f in full))
    assert dec[0][0].shape == (2, 17, 2) and np.isfinite(dec[0][0]).all()


def _full_size_properties(pkg, name, dtype, B, sub=(3, 5)):
    """Size-independent properties of a full-size forward + decode (the CPU oracle cannot run these batches):
    determinism, independence of batch position / batch size, well-formed decode."""
    import bench
    cfg = dict(bench.CONFIGS[name])
    model, codec, sd = bench.build(cfg, dtype, torch.device("cuda", 0))
    H, W = cfg["img"]
    x = pkg["syn"].synthetic_crops(B, H, W, seed=78).cuda()
    with torch.no_grad():
        full = [o.clone() for o in model(x)]
        again = model(x)
        for a, b in zip(full, again):
            assert torch.equal(a, b)
        lo, hi = sub
        part = model(x[lo:hi].contiguous())
        for a, b in zip(full, part):
            assert torch.equal(a[lo:hi], b)
        dec = codec.decode_device(full)
    k = dec["kpts"].cpu().numpy()
    assert k.shape == (B, cfg["K"], 2) and np.isfinite(k).all()
    assert (k[..., 0] >= -0.5 * W / (W // 4)).all() and (k[..., 0] <= W + 2).all()
    assert (k[..., 1] >= -0.5 * H / (H // 4)).all() and (k[..., 1] <= H + 2).all()
    hm = full[0]
    assert hm.shape == (B, cfg["K"], H // 4, W // 4) and float(hm.min()) >= 0 and float(hm.max()) <= 1
    del model
    torch.cuda.empty_cache()
    return cfg, sd, x, full


def test_full_size_forward_properties_vit_l_bs256_bf16(pkg):
    """BASELINE.json configs[2]: ViT-L 256x192 K=17 bf16, batch 256 (MFMA attention tiling)."""
    cfg, sd, x, full = _full_size_properties(pkg, "vit_l", torch.bfloat16, 256)
    with torch.no_grad():
        want = orc.model_forward(sd, x[:2].cpu(), patch=16, heads=cfg["heads"], pools=cfg["pools"])
    dh = (full[0][:2].cpu() - want[0]).abs()
    assert float(dh.mean()) < 0.03, float(dh.mean())


def test_full_size_forward_properties_vit_h_wholebody_bs128_bf16(pkg):
    """BASELINE.json configs[3]: ViT-H 384x288 K=133, 128 crops per GPU (x 8 GPUs = 1024 by the driver)."""
    cfg, sd, x, full = _full_size_properties(pkg, "vit_h_wholebody", torch.bfloat16, 128)
    with torch.no_grad():
        want = orc.model_forward(sd, x[:1].cpu(), patch=16, heads=cfg["heads"], pools=cfg["pools"])
    dh = (full[0][:1].cpu() - want[0]).abs()
    assert float(dh.mean()) < 0.03, float(dh.mean())


def test_full_size_forward_properties_vit_l_bs256_fp8(pkg):
    """BASELINE.json configs[4]: ViT-L fp8 weights / activations, batch 256.  No fp8 reference exists: properties
    + the deviation from the bf16 path of the same model on the same batch is bounded loosely and recorded."""
    import bench
    cfg, sd, x, full8 = _full_size_properties(pkg, "vit_l", torch.float8_e4m3fn, 256)
    model, codec, _ = bench.build(cfg, torch.bfloat16, torch.device("cuda", 0))
    with torch.no_grad():
        ref = model(x[:16].contiguous())
    d = (full8[0][:16].float() - ref[0].float()).abs()
    assert float(d.mean()) < 0.05, float(d.mean())
    _check_and_record_fraction("fp8_vit_l_heatmap_mean_abs_dev_vs_bf16_x1000", 1.0,
                               dict(heatmap_abs_mean=float(d.mean()), heatmap_abs_max=float(d.max())))


def test_run_inference_on_boxes_matches_crop_then_forward(pkg):
    """frame + boxes -> crops -> forward -> decode == forward of the Pillow-made crops; keypoints mapped
    back into the frame with the inverse of dataset.py:87-89."""
    from oracle import frontend_oracle as fo
    from probpose_pytorch_amd import inference
    model, hm_size = inference.build_model((192, 256), 17, "vit_s")
    from probpose_pytorch_amd.synthetic import synthetic_model_state
    model.load_state_dict(synthetic_model_state((256, 192), 16, 384, 12, 17, 3, (256, 256), seed=0))
    model = model.to("cuda").eval()
    codec = pkg["p"].Codec(pkg["p"].ProbMap((192, 256), hm_size, np.array([0.05] * 17)))
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    boxes = [(40.0, 30.0, 200.0, 380.0), (300.5, 100.25, 150.0, 300.0), (-20.0, 200.0, 180.0, 320.0)]
    out, preds, frame_kpts = inference.run_inference_on_boxes(model, codec, torch.from_numpy(frame).cuda(), boxes)
    crops = torch.from_numpy(np.stack([fo.scale_box_pil(frame, b, (192, 256)) for b in boxes])).cuda()
    out2, preds2 = inference.run_inference(model, codec, crops)
    assert torch.equal(out[0], out2[0])
    np.testing.assert_array_equal(preds[0][0], preds2[0][0])
    assert frame_kpts.shape == (3, 17, 2)
    np.testing.assert_allclose(frame_kpts[1], preds[0][0][1] / [192, 256] * [150.0, 300.0] + [300.5, 100.25])


@pytest.mark.parametrize("fp8_proj", [False, True])
def test_fp8_forward_tracks_bf16(pkg, fp8_proj, monkeypatch):
    """BASELINE config 5 (fp8 weights / activations on the fp8 MFMA): same model, bf16 vs fp8 ViT GEMMs.
    No reference exists for fp8 (the reference is fp32-only): the deviation from the bf16 path is bounded loosely and
    the decode of the fp8 heatmaps must stay well-formed; determinism and graph capture after calibration are checked."""
    from probpose_pytorch_amd import engine
    monkeypatch.setattr(engine, "FP8_PROJ", fp8_proj)
    model, _ = _build(pkg, (256, 192), 384, 4, 12, 17, [(4, 3), (2, 2), (2, 2)])
    x = pkg["syn"].synthetic_crops(8, 256, 192, seed=3).cuda()
    with torch.no_grad():
        model.set_compute_dtype(torch.bfloat16)
        ref = [t.float().clone() for t in model(x)]
        model.set_compute_dtype(torch.float8_e4m3fn)
        out = [t.float().clone() for t in model(x)]          # first call calibrates the activation scales
        again = [t.float().clone() for t in model(x)]
    for a, b in zip(out, again):
        assert torch.equal(a, b)
    d = (out[0] - ref[0]).abs()
    assert out[0].shape == ref[0].shape and torch.isfinite(out[0]).all()
    assert float(d.mean()) < 0.03 and float(d.max()) < 0.75, (float(d.mean()), float(d.max()))
    for a, b in zip(out[1:], ref[1:]):
        assert float((a - b).abs().max()) < 0.25
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        model(x)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g), torch.no_grad():
        cap = model(x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap[0].float(), out[0])


def test_optional_schedules_give_identical_results(pkg, monkeypatch):
    """engine.DUAL_CHAIN (two half-batch kernel chains on two streams) and engine.SERIALIZE_HEAD (no aux stream)
    only change the schedule: outputs must be bit-identical to the default path."""
    from probpose_pytorch_amd import engine
    model, _ = _build(pkg, (256, 192), 384, 3, 12, 17, [(4, 3), (2, 2), (2, 2)])
    model.set_compute_dtype(torch.bfloat16)
    x = pkg["syn"].synthetic_crops(16, 256, 192, seed=5).cuda()
    with torch.no_grad():
        ref = [t.clone() for t in model(x)]
        monkeypatch.setattr(engine, "DUAL_CHAIN", True)
        dual = [t.clone() for t in model(x)]
        monkeypatch.setattr(engine, "DUAL_CHAIN", False)
        monkeypatch.setattr(engine, "SERIALIZE_HEAD", True)
        serial = [t.clone() for t in model(x)]
    torch.cuda.synchronize()
    for a, b, c in zip(ref, dual, serial):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_timm_surface_returns_fresh_tensors(pkg):
    """timm's forward_features / forward return new tensors: two consecutive results must not alias the plan's
    cached workspace (in fp32 mode .float() is a no-op and a view of the scratch buffer would be overwritten)."""
    m, _ = _build(pkg, (64, 48), 128, 1, 2, 5, [(4, 3)], (64,))
    x1 = pkg["syn"].synthetic_crops(2, 64, 48, seed=1).cuda()
    x2 = pkg["syn"].synthetic_crops(2, 64, 48, seed=2).cuda()
    vit = m.backbone.model
    for dtype in (torch.float32, torch.bfloat16):
        m.set_compute_dtype(dtype)
        a = vit.forward_features(x1)
        a_copy = a.clone()
        b = vit.forward_features(x2)
        assert a.dtype == torch.float32 and a.shape == (2, 12, 128)
        assert torch.equal(a, a_copy) and not torch.equal(a, b)
        assert a.data_ptr() != b.data_ptr()
        assert torch.equal(vit(x1), a)
    # a smaller batch after a larger one reuses the leading rows of the same workspace; results unchanged
    m.set_compute_dtype(torch.float32)
    big = m(torch.cat([x1, x2]))
    small = m(x1)
    for u, v in zip(big, small):
        assert torch.equal(u[:2], v)


def test_fp8_explicit_calibration_covers_a_brighter_batch(pkg):
    """static fp8 activation scales came from the first batch with margin 1.0 and later batches with
    larger activations saturated silently.  calibrate_fp8() fixes the scales from given batches with head-room
    (margin 1.25); a model calibrated on both batches must track bf16 on the high-contrast batch at least as well as
    one calibrated on the flat batch only."""
    model, _ = _build(pkg, (256, 192), 384, 4, 12, 17, [(4, 3), (2, 2), (2, 2)])
    flat = (pkg["syn"].synthetic_crops(4, 256, 192, seed=3) * 0.2 + 0.4).cuda()
    g = torch.Generator().manual_seed(9)
    bright = (torch.rand((4, 3, 256, 192), generator=g) > 0.5).float().cuda()          # saturated black / white pixels
    with torch.no_grad():
        model.set_compute_dtype(torch.bfloat16)
        ref = model(bright)[0].float().clone()
        model.set_compute_dtype(torch.float8_e4m3fn)
        model.calibrate_fp8([flat])
        d_flat_only = float((model(bright)[0].float() - ref).abs().mean())
        model.calibrate_fp8([flat, bright])
        out = model(bright)[0].float()
        d_both = float((out - ref).abs().mean())
        again = model(bright)[0].float()
    assert torch.isfinite(out).all() and torch.equal(out, again)
    assert d_both <= d_flat_only * 1.05 + 1e-4, (d_both, d_flat_only)
    assert d_both < 0.05
    with pytest.raises(RuntimeError):
        model.set_compute_dtype(torch.bfloat16)
        model.calibrate_fp8([flat])


def test_checkpoint_interop_state_dict_roundtrip(pkg, tmp_path):
    """a checkpoint with the reference's parameter names (timm ViT names under
    backbone.model.*, head.* with BatchNorm running statistics and num_batches_tracked), written by torch.save outside
    probpose_pytorch_amd.synthetic, loaded through inference.load_weights (weights_only=True) and run in fp32 / bf16 /
    fp8: fp32 must match the CPU oracle evaluated on the same tensors to 1e-4, bf16 / fp8 stay within their bounds.
    Also the head-only checkpoint flavour of the reference CLI (inference.py:61-70, --model_type head)."""
    from probpose_pytorch_amd import inference
    img, C, depth, heads, K, pools = (256, 192), 384, 3, 6, 17, [(4, 3), (2, 2), (2, 2)]

    def fresh():
        return pkg["model"].ProbPoseModel(
            pkg["backbone"].ScratchViTBackbone(img, 16, embed_dim=C, depth=depth, num_heads=heads),
            pkg["head"].ProbMapHead(C, K, pools, (256, 256), (4, 4), final_layer_kernel_size=1))

    src = fresh()
    g = torch.Generator().manual_seed(2026)
    with torch.no_grad():
        for name, prm in src.named_parameters():
            if prm.ndim >= 2:
                fan_in = prm[0].numel() if "deconv" not in name else prm.shape[0] * prm.shape[2] * prm.shape[3]
                prm.copy_(torch.randn(prm.shape, generator=g) * fan_in ** -0.5)
            elif name.endswith("weight"):                       # LayerNorm / BatchNorm gamma
                prm.copy_(1.0 + 0.2 * torch.randn(prm.shape, generator=g))
            else:
                prm.copy_(0.1 * torch.randn(prm.shape, generator=g))
        for name, buf in src.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(0.3 * torch.randn(buf.shape, generator=g))
            elif name.endswith("running_var"):
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
            elif name.endswith("num_batches_tracked"):
                buf.fill_(1234)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    assert any(k.endswith("num_batches_tracked") for k in sd) and "backbone.model.blocks.0.attn.qkv.weight" in sd
    full_path, head_path = tmp_path / "full.pth", tmp_path / "head.pth"
    torch.save(sd, full_path)
    torch.save({k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}, head_path)

    x = pkg["syn"].synthetic_crops(3, *img, seed=8)
    with torch.no_grad():
        want = orc.model_forward(sd, x, patch=16, heads=heads, pools=pools)
    m = fresh()
    msg = inference.load_weights(m, full_path, "full")
    assert not msg.missing_keys and not msg.unexpected_keys
    m = m.cuda().eval()
    with torch.no_grad():
        got = m(x.cuda())
        for gt, wt, name in zip(got, want, ("heatmaps", "prob", "vis", "oks", "err")):
            np.testing.assert_allclose(gt.cpu().numpy(), wt.numpy(), rtol=0, atol=1e-4, err_msg=name)
        m.set_compute_dtype(torch.bfloat16)
        got16 = [t.float().cpu() for t in m(x.cuda())]
        m.set_compute_dtype(torch.float8_e4m3fn)
        got8 = [t.float().cpu() for t in m(x.cuda())]
    assert float((got16[0] - want[0]).abs().mean()) < 0.02
    assert float((got8[0] - got16[0]).abs().mean()) < 0.05 and torch.isfinite(got8[0]).all()
    # head-only checkpoint into a model whose backbone keeps its own weights
    m2 = fresh()
    m2.backbone.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")})
    msg = inference.load_weights(m2, head_path, "head")
    assert not msg.missing_keys and not msg.unexpected_keys
    m2 = m2.cuda().eval()
    with torch.no_grad():
        got2 = m2(x.cuda())
    for a, b in zip(got, got2):
        assert torch.equal(a, b)
    # a whole-module pickle is refused by the safe loader unless explicitly trusted
    torch.save(src, tmp_path / "module.pth")
    with pytest.raises(RuntimeError):
        inference.load_weights(fresh(), tmp_path / "module.pth", "full")


def test_fused_final_layer_is_bit_identical(pkg, monkeypatch):
    """engine.FUSE_FINAL: the last deconvolution's epilogue applies the final 1x1 layer (+ / T + clamp) instead of
    storing its 256-channel map for a separate launch.  Same bf16-rounded operands, same k order: the heatmaps must be
    bit-identical to the two-kernel path (and the aux outputs untouched)."""
    from probpose_pytorch_amd import engine
    for K, B in ((17, 5), (32, 2), (3, 1)):
        head = pkg["head"].ProbMapHead(384, K, [(4, 3), (2, 2), (2, 2)], (256, 256), (4, 4), final_layer_kernel_size=1)
        head.load_state_dict(pkg["syn"].synthetic_head_state(384, K, 3, (256, 256), seed=20 + K))
        head = head.cuda().eval().set_compute_dtype(torch.bfloat16)
        feats = pkg["syn"].synthetic_features(B, 384, 16, 12, seed=3).cuda()
        with torch.no_grad():
            monkeypatch.setattr(engine, "FUSE_FINAL", True)
            fused = [t.clone() for t in head(feats)]
            monkeypatch.setattr(engine, "FUSE_FINAL", False)
            plain = [t.clone() for t in head(feats)]
        assert fused[0].shape == (B, K, 64, 48) and float(fused[0].max()) > 0
        for a, b in zip(fused, plain):
            assert torch.equal(a, b)


def test_captured_graph_survives_many_other_batch_sizes(pkg):
    """A HIP graph captured at one batch size bakes in the ADDRESSES of the head's gather / row-map tables and of
    the workspace buffers.  Running more than eight other batch sizes eagerly afterwards (the table cache holds eight
    eager entries) must not free or reuse anything the graph reads: its replay still equals an eager forward."""
    m, _ = _build(pkg, (64, 48), 128, 2, 2, 17, [(4, 3)], deconv=(64, 64))
    m.set_compute_dtype(torch.bfloat16)
    xs = pkg["syn"].synthetic_crops(16, 64, 48, seed=5).cuda()
    x4 = xs[:4].clone()
    with torch.no_grad():
        m(x4)                                            # plans, tables, workspaces (eager)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(x4)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out_g = m(x4)
        g.replay()
        torch.cuda.synchronize()
        want = [o.clone() for o in m(x4)]
        for o, w in zip(out_g, want):
            assert torch.equal(o, w)
        junk = []
        for b in (1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 16):        # > 8 distinct eager batch sizes, some larger than 4
            m(xs[:b])
            junk.append(torch.full((1 << 18,), float(b), device="cuda"))   # churn the allocator over anything freed
        del junk
        torch.cuda.synchronize()
        for o in out_g:
            o.zero_()
        g.replay()
        torch.cuda.synchronize()
        for o, w in zip(out_g, want):
            assert torch.equal(o, w)
    tables = m.head._plan(xs.device)._tables
    assert sum(1 for v in tables.values() if v["pinned"]) >= 1
    assert sum(1 for v in tables.values() if not v["pinned"]) <= 8


def test_vit_b_bs64_fp32_every_crop_against_the_cpu_oracle(pkg):
    """The headline workload (ViT-B 256x192 K = 17, all 64 crops = 1 088 keypoints) in the exact-fp32 mode against the
    CPU oracle end to end.  Asserted without tolerance games: every output tensor within 1e-4 (the north_star's bound);
    for EVERY keypoint whose reference convolved map separates its two largest values by more than 1e-3 (a clear peak)
    the same integer peak, hence a keypoint within 1e-3 px; at least 99 % of those within 1e-4 px.  The remainder of the
    clear peaks (observed: 6 of ~1 000, up to 2.2e-4 px) are not flips: the sub-pixel step divides finite differences
    of the convolved map by its curvature (heatmap.py:160-161), which turns the <= 4e-5 heatmap deviation of a
    different float32 summation order into eps / |dxx| heatmap pixels x 4 input pixels on flat peaks -- the CPU path
    carries the same rounding noise against exact arithmetic.  Near-ties (plateaus of the clamp) are counted and each is
    verified to sit on a near-maximum of the reference map.  All counts are recorded."""
    import bench
    cfg = dict(bench.CONFIGS["vit_b"])
    model, codec, sd = bench.build(cfg, torch.float32, torch.device("cuda", 0))
    x = pkg["syn"].synthetic_crops(64, 256, 192, seed=1234)
    sig = bench.sigmas_for(17)
    with torch.no_grad():
        out = model(x.cuda())
        ref = orc.model_forward(sd, x, patch=16, heads=12, pools=cfg["pools"])
    for o, r in zip(out, ref):
        assert float((o.cpu() - r).abs().max()) <= 1e-4
    got = codec.decode(out)
    want = orc.codec_decode([r.numpy() for r in ref], (192, 256), (48, 64), sig)
    for a, b in zip(got[1:], want[1:]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    d = np.abs(got[0][0] - want[0][0]).max(-1)                      # (64, 17) px
    margin = np.empty_like(d)
    ref_heat = ref[0].numpy()
    for b in range(64):
        _, _, conv = orc.heatmap_expected_value(ref_heat[b], sig, "scipy", return_heatmap=True)
        top2 = np.sort(conv.reshape(17, -1), axis=1)[:, -2:]
        margin[b] = top2[:, 1] - top2[:, 0]
    clear = margin > 1e-3
    assert clear.sum() >= 0.5 * clear.size, f"only {clear.sum()} of {clear.size} keypoints have a clear peak"
    assert (d[clear] <= 1e-3).all(), f"a clear-peak keypoint moved by {d[clear].max():.3g} px: its arg-max flipped"
    assert (d[clear] <= 1e-4).mean() >= 0.99, f"only {(d[clear] <= 1e-4).mean():.4f} of the clear peaks within 1e-4 px"
    frac_all = _assert_keypoints_match_up_to_near_ties(pkg, out[0].cpu().numpy(), ref_heat, got[0][0], want[0][0], sig)
    _check_and_record_fraction("vit_b_bs64_fp32_all_crops", frac_all,
                               {"keypoints": int(d.size), "clear_peak": int(clear.sum()),
                                "clear_peak_within_1e-4": int((d[clear] <= 1e-4).sum()),
                                "near_tie": int((~clear).sum()), "near_tie_within_1e-4": int((d[~clear] <= 1e-4).sum()),
                                "max_px_clear": float(d[clear].max())})
