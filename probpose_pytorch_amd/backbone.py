"""Mirror of the reference's ``probpose/backbone.py``.

``ScratchViTBackbone`` keeps the reference constructor (backbone.py:24) and
exposes the same ``.model`` attribute tree timm's ``VisionTransformer`` has
(``patch_embed.proj``, ``pos_embed``, ``blocks[i].{norm1,attn.qkv,attn.proj,
norm2,mlp.fc1,mlp.fc2}``, ``norm``, ``forward_features``,
``patch_embed.dynamic_feat_size``), so state_dicts are interchangeable.  The
modules below only hold parameters; the arithmetic runs in the HIP kernels
through :mod:`probpose_pytorch_amd.engine`.  timm itself is not used.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib, engine, ops

_DEFAULT_DTYPE = torch.float32


class PatchEmbed(nn.Module):
    """timm ``PatchEmbed`` surface: Conv2d(3->C, k=p, s=p, bias) + flatten/transpose."""

    def __init__(self, img_size, patch_size: int, in_chans: int, embed_dim: int):
        super().__init__()
        self.img_size = (int(img_size[0]), int(img_size[1]))
        self.patch_size = (int(patch_size), int(patch_size))
        self.grid_size = (self.img_size[0] // patch_size, self.img_size[1] // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)

    def dynamic_feat_size(self, img_size):
        return img_size[0] // self.patch_size[0], img_size[1] // self.patch_size[1]


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class VisionTransformer(nn.Module):
    """Parameter container with timm==1.0.15 ``VisionTransformer`` names for
    ``num_classes=0, class_token=False, global_pool=''`` (no cls token, learned
    absolute pos-emb (1,N,C), pre-LN blocks, final ``norm``)."""

    def __init__(self, img_size=(256, 192), patch_size: int = 16, in_chans: int = 3, embed_dim: int = 384,
                 depth: int = 12, num_heads: int = 12, mlp_ratio: float = 4.0):
        super().__init__()
        if in_chans != 3:
            raise ValueError("the HIP patch-embed path is built for 3-channel crops")
        self.embed_dim = self.num_features = embed_dim
        self.num_heads = num_heads
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.compute_dtype = _DEFAULT_DTYPE
        self._init_weights()

    def _init_weights(self):
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def _plan(self, device):
        return engine.plan_for(self, engine.build_vit_plan, self.compute_dtype, device)

    @property
    def token_dtype(self) -> torch.dtype:
        """dtype of forward_tokens' result (fp8 mode keeps tokens, attention and the head in bf16)."""
        return torch.bfloat16 if self.compute_dtype == torch.float8_e4m3fn else self.compute_dtype

    def calibrate_fp8(self, batches, margin: float = None):
        """fp8 compute mode: fix the static activation scales from representative batches (see engine.VitPlan.calibrate)."""
        batches = list(batches)
        if not batches:
            raise ValueError("calibrate_fp8 needs at least one batch")
        _lib.require_device(batches[0])
        self._plan(batches[0].device).calibrate(batches, margin)
        return self

    def forward_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> device-resident tokens [B*N, C] in the compute dtype (HIP only)."""
        _lib.require_device(x)
        x = x.detach().contiguous().float()
        return self._plan(x.device).forward_tokens(x)

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> (B,N,C) float32, as timm's ``forward_features``."""
        B = x.shape[0]
        t = self.forward_tokens(x)
        # forward_tokens hands out the plan's cached workspace; timm returns a fresh tensor, so copy out (in fp32 mode
        # .float() alone would alias the scratch buffer and the next forward would overwrite this result)
        return t.to(torch.float32, copy=True).reshape(B, self.patch_embed.num_patches, self.embed_dim)

    def forward(self, x):
        return self.forward_features(x)


class ScratchViTBackbone(nn.Module):
    """Reference backbone.py:23-40.  ``input_image_size`` is (H, W).

    The reference hard-codes ``embed_dim=384`` with timm's default depth 12 /
    12 heads; ``embed_dim``/``depth``/``num_heads`` are extensions for the
    ViT-B/L/H configurations of BASELINE.json."""

    def __init__(self, input_image_size, patch_size: int, embed_dim: int = 384, depth: int = 12,
                 num_heads: int = 12, mlp_ratio: float = 4.0):
        super().__init__()
        self.model = VisionTransformer(img_size=input_image_size, patch_size=patch_size, embed_dim=embed_dim,
                                       depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio)

    def set_compute_dtype(self, dtype: torch.dtype):
        ops.dtype_code(dtype)
        self.model.compute_dtype = dtype
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> (B,C,gh,gw) float32 contiguous (backbone.py:35-40)."""
        B, _, height, width = x.shape
        tokens = self.model.forward_tokens(x)
        gh, gw = self.model.patch_embed.dynamic_feat_size((height, width))
        C = self.model.embed_dim
        out = torch.empty((B, C, gh, gw), dtype=torch.float32, device=x.device)
        ops.tokens_to_nchw(tokens, out, B, gh * gw, C)
        return out


class RadioBackbone(nn.Module):
    """Reference backbone.py:4-21 loads NVlabs/RADIO through ``torch.hub`` (a
    remote fetch of code and weights).  There is no network and no local copy,
    so this backbone is out of scope."""

    def __init__(self, version: str, mlp=None):
        super().__init__()
        raise NotImplementedError(
            "RadioBackbone needs torch.hub.load('NVlabs/RADIO', ...) (remote code + weights); "
            "only ScratchViTBackbone is built")
