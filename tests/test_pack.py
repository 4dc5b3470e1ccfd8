"""CPU tests of the host-side packing logic (BN folding, tap-major weights,
deconvolution parities, gather/scatter tables) against plain torch ops."""
import pytest
import torch
import torch.nn.functional as F

from probpose_pytorch_amd import pack


@pytest.mark.parametrize("k", [4, 3, 2])
def test_deconv_parities_reproduce_conv_transpose(k):
    torch.manual_seed(k)
    B, Cin, Cout, h, w = 2, 8, 6, 5, 4
    x, wt = torch.randn(B, Cin, h, w), torch.randn(Cin, Cout, k, k)
    pad, op = pack.deconv_geometry(k)
    ref = F.conv_transpose2d(x, wt, None, stride=2, padding=pad, output_padding=op)
    rows = x.permute(0, 2, 3, 1).reshape(B * h * w, Cin).contiguous()
    Wp = pack.pack_deconv_parities(wt, k)
    ro, rm = pack.deconv_tables(B, h, w, k, Cin)
    out = torch.zeros(B * 4 * h * w, Cout)
    seen = torch.zeros(B * 4 * h * w, dtype=torch.int32)
    for p in range(4):
        out[rm[p].long()] = pack.gather_rows(rows, ro[p], Cin) @ Wp[p].t()
        seen[rm[p].long()] += 1
    assert torch.all(seen == 1)                       # the four parities tile the output exactly once
    got = out.reshape(B, 2 * h, 2 * w, Cout).permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


def test_deconv_geometry_rejects_other_kernel_sizes():
    with pytest.raises(ValueError):
        pack.deconv_geometry(5)                        # reference head.py:452-457


@pytest.mark.parametrize("k,pad", [(3, 1), (1, 0), (5, 2)])
def test_conv_gather_reproduces_conv2d(k, pad):
    torch.manual_seed(0)
    x, wt = torch.randn(2, 8, 5, 4), torch.randn(6, 8, k, k)
    ref = F.conv2d(x, wt, None, padding=pad)
    rows = x.permute(0, 2, 3, 1).reshape(-1, 8).contiguous()
    A = pack.gather_rows(rows, pack.conv_gather_table(2, 5, 4, k, k, pad, pad, 8), 8)
    got = (A @ pack.conv_taps_major(wt).t()).reshape(2, 5, 4, 6).permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


def test_fold_bn_matches_eval_batchnorm():
    torch.manual_seed(1)
    x = torch.randn(2, 8, 5, 4)
    conv = torch.nn.Conv2d(8, 6, 3, padding=1)
    bn = torch.nn.BatchNorm2d(6).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    wf, bf = pack.fold_bn(conv.weight.detach(), conv.bias.detach(), bn.weight.detach(), bn.bias.detach(),
                          bn.running_mean, bn.running_var, bn.eps)
    with torch.no_grad():
        torch.testing.assert_close(F.conv2d(x, wf, bf, padding=1), bn(conv(x)), rtol=1e-5, atol=1e-5)


def test_module_state_dicts_are_interchangeable_with_reference_names():
    """The HIP-backed modules keep the reference / timm parameter names."""
    from probpose_pytorch_amd.backbone import ScratchViTBackbone
    from probpose_pytorch_amd.head import ProbMapHead
    from probpose_pytorch_amd.model import ProbPoseModel
    from probpose_pytorch_amd.synthetic import synthetic_model_state
    m = ProbPoseModel(ScratchViTBackbone((64, 48), 16, embed_dim=128, depth=2, num_heads=2),
                      ProbMapHead(128, 17, [(4, 3)], (64, 64), (4, 4)))
    sd = synthetic_model_state((64, 48), 16, 128, 2, 17, 1, (64, 64))
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    names = dict(m.named_modules())
    for n in ("backbone.model.patch_embed.proj", "backbone.model.blocks.1.attn.qkv", "backbone.model.blocks.0.mlp.fc2",
              "backbone.model.norm", "head.deconv_layers.0", "head.deconv_layers.1", "head.final_layer",
              "head.probability_layers.0", "head.visibility_layers.4", "head.oks_layers.1", "head.error_layers.4"):
        assert n in names
    assert m.head.temperature == 0.5 and isinstance(m.head.normalize_layer, torch.nn.Identity)
    assert m.backbone.model.patch_embed.dynamic_feat_size((64, 48)) == (4, 3)
    with pytest.raises(ValueError):
        ProbMapHead(8, 2, [(2, 2)], (4, 4), (4,))     # mismatched lengths, reference head.py:188-196


def test_modules_refuse_cpu_tensors():
    from probpose_pytorch_amd import _lib
    from probpose_pytorch_amd.backbone import ScratchViTBackbone
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    bb = ScratchViTBackbone((64, 48), 16, embed_dim=128, depth=1, num_heads=2)
    with pytest.raises(_lib.HipExtensionError):
        bb(torch.zeros(1, 3, 64, 48))


def test_drop_in_import_paths():
    """`import probpose.<module>` (the reference's package name) resolves to the HIP-backed modules."""
    import probpose.codec
    import probpose_pytorch_amd.codec
    from probpose.backbone import ScratchViTBackbone  # noqa: F401
    from probpose.codec import Codec, ProbMap  # noqa: F401
    from probpose.head import ProbMapHead  # noqa: F401
    from probpose.heatmap import get_heatmap_expected_value  # noqa: F401
    from probpose.model import ProbPoseModel  # noqa: F401
    from probpose.util import to_numpy  # noqa: F401
    assert probpose.codec is probpose_pytorch_amd.codec


def test_inference_harness_host_logic(tmp_path):
    """Counterpart of the reference CLI (inference.py:61-112): model construction for an input size,
    safe checkpoint loading, pool schedule that always ends at 1x1."""
    from probpose_pytorch_amd import inference
    for grid in [(16, 12), (24, 18), (24, 24), (8, 8)]:
        h, w = grid
        for p in inference.default_pools(grid):
            h, w = h // p[0], w // p[1]
        assert (h, w) == (1, 1)
    model, hm_size = inference.build_model((192, 256), 17, "vit_s")
    assert hm_size == (48, 64) and model.backbone.model.embed_dim == 384
    torch.save(model.state_dict(), tmp_path / "full.pt")
    res = inference.load_weights(model, tmp_path / "full.pt", "full")
    assert not res.missing_keys
    torch.save(model, tmp_path / "pickled_module.pt")           # the reference's torch.save(model) style
    with pytest.raises(RuntimeError, match="state_dict"):
        inference.load_weights(model, tmp_path / "pickled_module.pt", "full")
