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
    assert frac >= 0.7


@pytest.mark.parametrize("cfg", [
    dict(img=(64, 48), C=128, depth=2, heads=2, K=17, pools=[(4, 3)], deconv=(64, 64), B=3),
    dict(img=(256, 192), C=384, depth=12, heads=12, K=17, pools=[(4, 3), (2, 2), (2, 2)], deconv=(256, 256), B=1),
    dict(img=(256, 192), C=768, depth=12, heads=12, K=17, pools=[(4, 3), (2, 2), (2, 2)], deconv=(256, 256), B=2),
], ids=["tiny", "S1-vit-s-256x192", "vit-b-256x192"])
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
    codec = pkg["p"].Codec(pkg["p"].ProbMap((W, H), (W // 4, H // 4), orc.COCO17_SIGMAS))
    dec = codec.decode(got)
    ref = orc.codec_decode([w.numpy() for w in want], (W, H), (W // 4, H // 4), orc.COCO17_SIGMAS)
    for a, b in zip(dec[1:4], ref[1:4]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    np.testing.assert_allclose(dec[4], ref[4], rtol=0, atol=1e-4)
    frac = _assert_keypoints_match_up_to_near_ties(pkg, got[0].cpu().numpy(), want[0].numpy(), dec[0][0], ref[0][0],
                                                   orc.COCO17_SIGMAS)
    print(f"\n[{cfg['C']}] keypoints within 1e-4 px of the CPU path: {frac:.2%} (the rest are verified near-ties)")
    assert frac >= 0.7


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
