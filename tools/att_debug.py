#!/usr/bin/env python3
"""Diagnostic: rows of pp_attention's output that differ from torch's SDPA, for a list of shapes (bf16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from probpose_pytorch_amd import _lib, ops
if len(sys.argv) > 1:      # lab library instead of the shipped one
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "probpose_pytorch_amd", "lib", "exp", sys.argv[1])

for B, N, heads, hd in [(2, 200, 2, 32), (1, 208, 2, 32)]:
    C = heads * hd
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn((B * N, 3 * C), generator=g).to(torch.bfloat16).cuda()
    out = torch.full((B * N, C), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.attention(qkv, out, B, N, heads, hd)
    q, k, v = qkv.double().reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * N, C)
    bad = ~((out.double() - ref).abs() <= 0.02 + 2 ** -6 * ref.abs())
    rows = bad.any(dim=1).nonzero().flatten().tolist()
    nan_rows = torch.isnan(out.float()).any(dim=1).nonzero().flatten().tolist()
    print((B, N, heads, hd), "bad rows:", len(rows), rows[:40], "| nan rows", len(nan_rows), "| bad cols of first bad row:",
          bad[rows[0]].nonzero().flatten().tolist()[:8] if rows else [], "| values:", out[rows[0], :8].float().tolist() if rows else [])
