"""Host-side weight packing and gather-table construction (pure torch, CPU or
GPU, no HIP): BatchNorm folding, tap-major conv weights, the four stride-2
deconvolution parities, and the row-gather / row-scatter tables that let one
MFMA GEMM kernel run every convolution of the head as an implicit GEMM.

All activations on the device are channels-last rows: a feature map
(B, C, h, w) is stored as [B*h*w, C] (row = (b*h + y)*w + x), which is exactly
the (B, N, C) token layout the ViT emits, so the reference's
permute(0,3,1,2).contiguous() (backbone.py:40) never has to happen between
backbone and head.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def fold_bn(w: torch.Tensor, conv_bias: Optional[torch.Tensor], bn_w, bn_b, mean, var,
            eps: float, out_dim: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fold an eval-mode BatchNorm2d into the preceding (de)convolution.

    y = (conv(x) + b - mean) * gamma / sqrt(var + eps) + beta
    ``out_dim`` is the output-channel dim of ``w`` (0 for Conv2d, 1 for ConvTranspose2d).
    """
    scale = bn_w.double() / torch.sqrt(var.double() + eps)
    shape = [1] * w.dim()
    shape[out_dim] = -1
    wf = (w.double() * scale.reshape(shape)).float()
    b0 = conv_bias.double() if conv_bias is not None else torch.zeros_like(scale)
    bf = ((b0 - mean.double()) * scale + bn_b.double()).float()
    return wf, bf


def conv_taps_major(w: torch.Tensor) -> torch.Tensor:
    """Conv2d weight (Cout, Cin, kh, kw) -> (Cout, kh*kw*Cin), k = (ky*kw + kx)*Cin + ci."""
    Cout, Cin, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(Cout, kh * kw * Cin).contiguous()


def conv_gather_table(B: int, h: int, w: int, kh: int, kw: int, ph: int, pw: int,
                      row_stride: int) -> torch.Tensor:
    """int32 [kh*kw, B*h*w]: element offset of input row (b, y+ky-ph, x+kx-pw) or -1 (zero pad)."""
    b = torch.arange(B).view(B, 1, 1)
    y = torch.arange(h).view(1, h, 1)
    x = torch.arange(w).view(1, 1, w)
    taps = []
    for ky in range(kh):
        for kx in range(kw):
            yy, xx = y + ky - ph, x + kx - pw
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            off = ((b * h + yy) * w + xx) * row_stride
            taps.append(torch.where(ok.expand(B, h, w), off.expand(B, h, w), torch.full((), -1)).reshape(-1))
    t = torch.stack(taps)
    assert t.max().item() < 2 ** 31
    return t.to(torch.int32).contiguous()


def deconv_axis_taps(k: int, pad: int) -> List[List[Tuple[int, Optional[int]]]]:
    """Stride-2 transposed convolution along one axis, split by output parity.

    Output o = 2*i + par receives input i+delta through kernel index
    kidx = par + pad - 2*delta (from o = 2*iy - pad + kidx).  Returns, for
    par in (0,1), exactly two (delta, kidx) taps, padded with (0, None)."""
    out = []
    for par in (0, 1):
        taps = []
        for delta in (0, -1, 1):
            kidx = par + pad - 2 * delta
            if 0 <= kidx < k:
                taps.append((delta, kidx))
        assert 1 <= len(taps) <= 2
        while len(taps) < 2:
            taps.append((0, None))
        out.append(taps)
    return out


def deconv_geometry(k: int) -> Tuple[int, int]:
    """(padding, output_padding) the reference picks per kernel size (head.py:443-451)."""
    if k == 4:
        return 1, 0
    if k == 3:
        return 1, 1
    if k == 2:
        return 0, 0
    raise ValueError(f"Unsupported kernel size {k} for deconvlutional layers in ProbMapHead")


def pack_deconv_parities(w: torch.Tensor, k: int) -> torch.Tensor:
    """ConvTranspose2d weight (Cin, Cout, k, k) [BN already folded on dim 1] ->
    (4, Cout, 4*Cin): parity p = py*2+px, K index = (a*2+b)*Cin + ci for axis taps a (y), b (x)."""
    Cin, Cout = w.shape[0], w.shape[1]
    pad, _ = deconv_geometry(k)
    taps = deconv_axis_taps(k, pad)
    out = torch.zeros((4, Cout, 4 * Cin), dtype=w.dtype)
    for py in (0, 1):
        for px in (0, 1):
            for a, (_, ky) in enumerate(taps[py]):
                for b, (_, kx) in enumerate(taps[px]):
                    if ky is None or kx is None:
                        continue
                    t = a * 2 + b
                    out[py * 2 + px, :, t * Cin:(t + 1) * Cin] = w[:, :, ky, kx].t()
    return out


def deconv_tables(B: int, h: int, w: int, k: int, row_stride: int):
    """Gather / scatter tables of one stride-2 deconvolution (h,w) -> (2h,2w).

    rowoff int32 [4 parities, 4 taps, B*h*w] (element offsets, -1 = zero),
    rowmap int32 [4 parities, B*h*w] (output row of GEMM row m)."""
    pad, _ = deconv_geometry(k)
    taps = deconv_axis_taps(k, pad)
    b = torch.arange(B).view(B, 1, 1)
    i = torch.arange(h).view(1, h, 1)
    j = torch.arange(w).view(1, 1, w)
    rowoff, rowmap = [], []
    for py in (0, 1):
        for px in (0, 1):
            per_tap = []
            for (dy, ky) in taps[py]:
                for (dx, kx) in taps[px]:
                    yy, xx = i + dy, j + dx
                    ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                    if ky is None or kx is None:
                        ok = ok & False
                    off = ((b * h + yy) * w + xx) * row_stride
                    per_tap.append(torch.where(ok.expand(B, h, w), off.expand(B, h, w),
                                               torch.full((), -1)).reshape(-1))
            rowoff.append(torch.stack(per_tap))
            rowmap.append((((b * 2 * h + 2 * i + py) * (2 * w)) + 2 * j + px).expand(B, h, w).reshape(-1))
    ro, rm = torch.stack(rowoff), torch.stack(rowmap)
    assert ro.max().item() < 2 ** 31 and rm.max().item() < 2 ** 31
    return ro.to(torch.int32).contiguous(), rm.to(torch.int32).contiguous()


def gather_rows(A: torch.Tensor, rowoff: torch.Tensor, seg_len: int) -> torch.Tensor:
    """Reference semantics of the GEMM's A addressing (used by CPU tests):
    A(m, k) = A.flat[rowoff[k // seg_len, m] + k % seg_len], 0 where rowoff < 0."""
    flat = A.reshape(-1)
    segs, M = rowoff.shape
    cols = torch.arange(seg_len)
    out = torch.zeros((M, segs * seg_len), dtype=A.dtype)
    for s in range(segs):
        off = rowoff[s].long()
        ok = off >= 0
        idx = off.clamp(min=0)[:, None] + cols[None]
        out[:, s * seg_len:(s + 1) * seg_len] = torch.where(ok[:, None], flat[idx], torch.zeros((), dtype=A.dtype))
    return out


def pool_out(h: int, w: int, k) -> Tuple[int, int, int, int]:
    kh, kw = (k, k) if isinstance(k, int) else (int(k[0]), int(k[1]))
    return kh, kw, h // kh, w // kw
