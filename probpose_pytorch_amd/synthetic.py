"""Seeded synthetic weights and inputs (no network: there are no checkpoints or
datasets to fetch).  Used by bench.py, the tests and the golden generator.

The reference's own initialisation (head.py:476-485: N(0, 1e-3) conv weights)
produces heatmaps of ~1e-4 and sigmoids pinned at 0.5, which would make every
parity check vacuous; this recipe (with the affine terms
randomised as well so bias / gamma / beta paths are exercised) keeps the
activations O(1) through every layer.

This module depends on torch only, so the golden generator can load it by path
next to the reference checkout.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch


def _gen(seed: int) -> torch.Generator:
    return torch.Generator().manual_seed(int(seed))


def _normal(g, shape, std):
    return torch.randn(shape, generator=g, dtype=torch.float32) * std


def _bn(g, sd, prefix, C):
    sd[prefix + "weight"] = 0.8 + 0.4 * torch.rand(C, generator=g)
    sd[prefix + "bias"] = _normal(g, (C,), 0.1)
    sd[prefix + "running_mean"] = _normal(g, (C,), 0.1)
    sd[prefix + "running_var"] = 0.5 + torch.rand(C, generator=g)
    sd[prefix + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def synthetic_head_state(C: int, K: int, n_pools: int = 3, deconv_out=(256, 256), seed: int = 0,
                         final_kernel=1, conv_out=(), conv_kernels=()) -> "OrderedDict[str, torch.Tensor]":
    """state_dict of ``ProbMapHead(C, K, pools, deconv_out, (4,)*n, conv_out, conv_kernels,
    final_layer_kernel_size=final_kernel)`` (parameter names of reference head.py: deconv_layers / conv_layers /
    final_layer / *_layers; ``final_kernel=None``: nn.Identity, no final parameters)."""
    g = _gen(seed)
    sd = OrderedDict()
    cin = C
    for i, cout in enumerate(deconv_out):
        # ConvTranspose2d weight (Cin, Cout, 4, 4); every output pixel sees 2x2 taps
        sd[f"deconv_layers.{3 * i}.weight"] = _normal(g, (cin, cout, 4, 4), math.sqrt(2.0 / (cin * 4)))
        _bn(g, sd, f"deconv_layers.{3 * i + 1}.", cout)
        cin = cout
    for i, (cout, k) in enumerate(zip(conv_out, conv_kernels)):
        sd[f"conv_layers.{3 * i}.weight"] = _normal(g, (cout, cin, k, k), math.sqrt(2.0 / (cin * k * k)))
        sd[f"conv_layers.{3 * i}.bias"] = _normal(g, (cout,), 0.1)
        _bn(g, sd, f"conv_layers.{3 * i + 1}.", cout)
        cin = cout
    if final_kernel is not None:
        sd["final_layer.weight"] = _normal(g, (K, cin, final_kernel, final_kernel),
                                           math.sqrt(1.0 / (cin * final_kernel ** 2)))
        sd["final_layer.bias"] = _normal(g, (K,), 0.1)
    for name in ("probability", "visibility", "oks", "error"):
        for i in range(n_pools):
            sd[f"{name}_layers.{4 * i}.weight"] = _normal(g, (C, C, 3, 3), math.sqrt(2.0 / (C * 9)))
            sd[f"{name}_layers.{4 * i}.bias"] = _normal(g, (C,), 0.1)
            _bn(g, sd, f"{name}_layers.{4 * i + 1}.", C)
        sd[f"{name}_layers.{4 * n_pools}.weight"] = _normal(g, (K, C, 1, 1), math.sqrt(1.0 / C))
        sd[f"{name}_layers.{4 * n_pools}.bias"] = _normal(g, (K,), 0.1)
    return sd


def synthetic_vit_state(img_size=(256, 192), patch: int = 16, embed_dim: int = 384, depth: int = 12,
                        mlp_ratio: float = 4.0, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """state_dict with timm ``VisionTransformer`` names (class_token=False, num_classes=0)."""
    g = _gen(seed)
    C = embed_dim
    N = (img_size[0] // patch) * (img_size[1] // patch)
    hidden = int(C * mlp_ratio)
    sd = OrderedDict()
    sd["pos_embed"] = _normal(g, (1, N, C), 0.02)
    sd["patch_embed.proj.weight"] = _normal(g, (C, 3, patch, patch), math.sqrt(1.0 / (3 * patch * patch)))
    sd["patch_embed.proj.bias"] = _normal(g, (C,), 0.02)

    def ln(prefix):
        sd[prefix + "weight"] = 0.9 + 0.2 * torch.rand(C, generator=g)
        sd[prefix + "bias"] = _normal(g, (C,), 0.05)

    def lin(prefix, out_f, in_f):
        sd[prefix + "weight"] = _normal(g, (out_f, in_f), math.sqrt(1.0 / in_f))
        sd[prefix + "bias"] = _normal(g, (out_f,), 0.02)

    for i in range(depth):
        p = f"blocks.{i}."
        ln(p + "norm1.")
        lin(p + "attn.qkv.", 3 * C, C)
        lin(p + "attn.proj.", C, C)
        ln(p + "norm2.")
        lin(p + "mlp.fc1.", hidden, C)
        lin(p + "mlp.fc2.", C, hidden)
    ln("norm.")
    return sd


def synthetic_model_state(img_size=(256, 192), patch=16, embed_dim=768, depth=12, K=17, n_pools=3,
                          deconv_out=(256, 256), seed: int = 0):
    """state_dict of ``ProbPoseModel(ScratchViTBackbone, ProbMapHead)``."""
    sd = OrderedDict()
    for k, v in synthetic_vit_state(img_size, patch, embed_dim, depth, seed=seed).items():
        sd["backbone.model." + k] = v
    for k, v in synthetic_head_state(embed_dim, K, n_pools, deconv_out, seed=seed + 1).items():
        sd["head." + k] = v
    return sd


def synthetic_crops(B: int, H: int, W: int, seed: int = 1234) -> torch.Tensor:
    """Person crops as the reference feeds them: fp32 in [0,1), no mean/std
    (inference.py:76-82)."""
    return torch.rand((B, 3, H, W), generator=_gen(seed), dtype=torch.float32)


def synthetic_features(B: int, C: int, h: int, w: int, seed: int = 0) -> torch.Tensor:
    return torch.randn((B, C, h, w), generator=_gen(seed), dtype=torch.float32)
