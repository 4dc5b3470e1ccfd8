#!/usr/bin/env python3
"""Time pp_attention (bf16) on the BASELINE configs' shapes with HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

g.build()
from probpose_pytorch_amd import ops


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


for name, B, N, heads, hd in [("vit_b bs64", 64, 192, 12, 64), ("vit_l bs256", 256, 192, 16, 64),
                              ("vit_s bs64", 64, 192, 12, 32), ("vit_h 384x288 bs128", 128, 432, 16, 80),
                              ("vit_l 384x288 bs64", 64, 432, 16, 64)]:
    C = heads * hd
    qkv = torch.randn((B * N, 3 * C), generator=torch.Generator().manual_seed(0)).to(torch.bfloat16).cuda()
    out = torch.empty((B * N, C), dtype=torch.bfloat16, device="cuda")
    t = bench(lambda: ops.attention(qkv, out, B, N, heads, hd))
    fl = 4.0 * N * N * hd * heads * B
    print(f"{name:24s} N={N} hd={hd}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s", flush=True)
