"""Mirror of the reference's ``probpose/model.py``."""
import torch
from torch import Tensor, nn

from . import ops
from .backbone import ScratchViTBackbone
from .head import ProbMapHead


class ProbPoseModel(nn.Module):
    """Reference model.py:4-11: ``forward(x) = head(backbone(x))``.

    When both halves are the HIP-backed modules of this package the feature map
    never leaves the channels-last token layout the ViT produces (the
    reference's permute+contiguous copy, backbone.py:40, is skipped)."""

    def __init__(self, backbone, head):
        super().__init__()
        self.backbone = backbone
        self.head = head

    def set_compute_dtype(self, dtype: torch.dtype):
        """torch.float32 (exact-fp32 MFMA, parity mode; default), torch.bfloat16, or torch.float8_e4m3fn:
        the ViT's qkv / fc1 / fc2 GEMMs on fp8 MFMA (e4m3 weights with per-output-channel scales, e4m3
        activations with static per-tensor scales calibrated on the first batch), everything else bf16."""
        self.backbone.set_compute_dtype(dtype)
        self.head.set_compute_dtype(torch.bfloat16 if dtype == torch.float8_e4m3fn else dtype)
        return self

    def calibrate_fp8(self, batches, margin: float = None):
        """fp8 mode only: static activation scales = margin (default 1.25) x the maximum |activation| over the given
        batches / 448.  Call after set_compute_dtype(torch.float8_e4m3fn); otherwise the first forward calibrates on
        its own batch."""
        self.backbone.model.calibrate_fp8(batches, margin)
        return self

    def forward(self, x: Tensor):
        if isinstance(self.backbone, ScratchViTBackbone) and isinstance(self.head, ProbMapHead) \
                and self.backbone.model.token_dtype == self.head.compute_dtype:
            B, _, height, width = x.shape
            tokens = self.backbone.model.forward_tokens(x)
            gh, gw = self.backbone.model.patch_embed.dynamic_feat_size((height, width))
            return self.head.forward_tokens(tokens, B, gh, gw)
        return self.head(self.backbone(x))
