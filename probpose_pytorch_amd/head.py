"""Mirror of the reference's ``probpose/head.py`` (``ProbMapHead``).

Same constructor signature, sub-module names and parameter layout as the
reference (head.py:83-167), so state_dicts are interchangeable; the forward
runs in HIP (eval-mode BatchNorm folded into the preceding convolution, the
stride-2 deconvolutions and 3x3 convolutions as implicit MFMA GEMMs, the final
1x1 + /temperature + clamp fused into the last GEMM's epilogue).
"""
from __future__ import annotations

from typing import Sequence, Tuple, Union

import torch
from torch import Tensor, nn

from . import _lib, engine, ops
from .pack import deconv_geometry


class Sparsemax(nn.Module):
    """Stand-in for ``sparsemax.Sparsemax(dim=-1)`` (head.py:11,241; third-party sparsemax==0.1.9, not part of the
    reference checkout): holds no parameters, so state_dicts are unchanged.  The HIP head applies the projection
    itself (``pp_sparsemax_rows``: Martins & Astudillo 2016, parity unpinned); calling the module directly runs the
    same kernel on a contiguous float32 GPU tensor."""

    def __init__(self, dim: int = -1):
        super().__init__()
        if dim != -1:
            raise ValueError("the HIP Sparsemax normalises the last axis (the reference uses dim=-1, head.py:241)")
        self.dim = dim

    def forward(self, x: Tensor) -> Tensor:
        _lib.require_device(x)
        out = x.detach().float().contiguous().clone()
        return ops.sparsemax_rows(out, 1.0)      # sparsemax lies in [0, 1]: with scale 1 the kernel's clamp is idle


class ProbMapHead(nn.Module):
    def __init__(
        self,
        in_channels: Union[int, Sequence[int]],
        out_channels: int,
        alt_head_kernel_sizes: Sequence,
        deconv_out_channels=(256, 256, 256),
        deconv_kernel_sizes=(4, 4, 4),
        conv_out_channels=None,
        conv_kernel_sizes=None,
        final_layer_kernel_size=1,
        normalize: float | None = None,
        detach_probability: bool = True,
        detach_visibility: bool = True,
        freeze_heatmaps: bool = False,
        freeze_probability: bool = False,
        freeze_visibility: bool = False,
        freeze_oks: bool = False,
        freeze_error: bool = False,
    ):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.temperature = 0.5                       # head.py:107
        self.nonlinearity = nn.ReLU(inplace=True)
        self.normalize = normalize
        self.detach_probability = detach_probability
        self.detach_visibility = detach_visibility
        self.freeze_oks = freeze_oks
        self.freeze_error = freeze_error
        self.compute_dtype = torch.float32

        # ---- heatmap branch (head.py:174-253)
        c = in_channels
        if deconv_out_channels:
            if deconv_kernel_sizes is None or len(deconv_out_channels) != len(deconv_kernel_sizes):
                raise ValueError('"deconv_out_channels" and "deconv_kernel_sizes" should be integer sequences '
                                 f"with the same length. Got mismatched lengths {deconv_out_channels} and "
                                 f"{deconv_kernel_sizes}")
            mods = []
            for oc, k in zip(deconv_out_channels, deconv_kernel_sizes):
                pad, opad = deconv_geometry(k)
                mods += [nn.ConvTranspose2d(c, oc, kernel_size=k, stride=2, padding=pad, output_padding=opad,
                                            bias=False), nn.BatchNorm2d(oc), self.nonlinearity]
                c = oc
            self.deconv_layers = nn.Sequential(*mods)
        else:
            self.deconv_layers = nn.Identity()
        if conv_out_channels:
            if conv_kernel_sizes is None or len(conv_out_channels) != len(conv_kernel_sizes):
                raise ValueError('"conv_out_channels" and "conv_kernel_sizes" should be integer sequences '
                                 f"with the same length. Got mismatched lengths {conv_out_channels} and "
                                 f"{conv_kernel_sizes}")
            mods = []
            for oc, k in zip(conv_out_channels, conv_kernel_sizes):
                mods += [nn.Conv2d(c, oc, kernel_size=k, stride=1, padding=(k - 1) // 2), nn.BatchNorm2d(oc),
                         self.nonlinearity]
                c = oc
            self.conv_layers = nn.Sequential(*mods)
        else:
            self.conv_layers = nn.Identity()
        if final_layer_kernel_size is not None:
            self.final_layer = nn.Conv2d(c, out_channels, kernel_size=final_layer_kernel_size,
                                         padding=final_layer_kernel_size // 2)
        else:
            self.final_layer = nn.Identity()
        if normalize is None:
            self.normalize_layer = nn.Identity()
        else:
            self.normalize_layer = Sparsemax(dim=-1)

        # ---- aux branches (head.py:255-405)
        def aux(last):
            mods = []
            for ks in alt_head_kernel_sizes:
                mods += [nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1),
                         nn.BatchNorm2d(in_channels), nn.MaxPool2d(kernel_size=ks, stride=ks, padding=0),
                         self.nonlinearity]
            mods += [nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0), last]
            return nn.Sequential(*mods)

        self.probability_layers = aux(nn.Sigmoid())
        self.visibility_layers = aux(nn.Sigmoid())
        self.oks_layers = aux(nn.Sigmoid())
        self.error_layers = aux(self.nonlinearity)
        for flag, mods in ((freeze_heatmaps, (self.deconv_layers, self.conv_layers, self.final_layer)),
                           (freeze_probability, (self.probability_layers,)),
                           (freeze_visibility, (self.visibility_layers,)), (freeze_oks, (self.oks_layers,)),
                           (freeze_error, (self.error_layers,))):
            if flag:
                for m in mods:
                    for p in m.parameters():
                        p.requires_grad = False
        self._initialize_weights()
        self.eval()          # the HIP path implements eval-mode BatchNorm (running statistics)

    def _initialize_weights(self):
        """head.py:476-485."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def set_compute_dtype(self, dtype: torch.dtype):
        ops.dtype_code(dtype)
        self.compute_dtype = dtype
        return self

    # ------------------------------------------------------------------ HIP
    def _plan(self, device):
        if self.training:
            raise RuntimeError("ProbMapHead runs eval-mode BatchNorm on the HIP path; call .eval() "
                               "(the reference CLI's train-mode BN is a bug, SURVEY.md section 3.2)")
        return engine.plan_for(self, engine.build_head_plan, self.compute_dtype, device)

    def forward_tokens(self, tokens: Tensor, B: int, h: int, w: int):
        """Channels-last rows [B*h*w, C] (compute dtype) -> the reference 5-tuple."""
        return self._plan(tokens.device).forward(tokens, B, h, w)

    def _to_tokens(self, x: Tensor):
        _lib.require_device(x)
        B, C, h, w = x.shape
        x = x.detach().contiguous().float()
        tokens = torch.empty((B * h * w, C), dtype=self.compute_dtype, device=x.device)
        ops.nchw_to_tokens(x, tokens, B, C, h * w)
        return tokens, B, h, w

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
        """(B,C,h,w) -> (heatmaps, probabilities, visibilities, oks, error).  head.py:487-511."""
        return self.forward_tokens(*self._to_tokens(x))

    def forward_heatmap(self, x: Tensor) -> Tensor:
        return self.forward(x)[0]

    def forward_probability(self, x: Tensor) -> Tensor:
        return self.forward(x)[1]

    def forward_visibility(self, x: Tensor) -> Tensor:
        return self.forward(x)[2]

    def forward_oks(self, x: Tensor) -> Tensor:
        return self.forward(x)[3]

    def forward_error(self, x: Tensor) -> Tensor:
        return self.forward(x)[4]
