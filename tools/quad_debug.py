#!/usr/bin/env python3
"""Diagnostic for the four-wave GEMM forms: where (16 x 16 block map) a tile's result differs from the fp64 product."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from probpose_pytorch_amd import ops

M, N, K, tile = (int(v) for v in sys.argv[1:5])
g = torch.Generator().manual_seed(1)
A = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
W = (torch.randn((N, K), generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
BIAS, GELU = "bias" in sys.argv, "gelu" in sys.argv
b = torch.randn((N,), generator=g).cuda() if BIAS else None
pre = A.double() @ W.double().t() + (b.double() if BIAS else 0)
ref = (torch.nn.functional.gelu(pre) if GELU else pre).cpu().numpy()
for rep in range(3):
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.linear(A, W, b, out=out, tile=tile, epilogue=ops.EPI_GELU if GELU else 0)
    o = out.double().cpu().numpy()
    bad = ~(np.abs(o - ref) <= 0.02 + 2 ** -7 * np.abs(ref))
    print(f"rep {rep}: {bad.sum()} bad of {bad.size}; nan {np.isnan(o).sum()}")
    if bad.any():
        rows, cols = np.nonzero(bad)
        print("  rows:", np.unique(rows)[:40], "...", "cols:", np.unique(cols)[:40])
        bm = bad[: M // 16 * 16, : N // 16 * 16].reshape(M // 16, 16, N // 16, 16).sum(axis=(1, 3))
        rb, cb = np.nonzero(bm)
        print("  16x16 blocks (row blk, col blk, count):", [(int(r), int(c), int(bm[r, c])) for r, c in zip(rb, cb)][:40])
        r, c = rows[0], cols[0]
        print("  e.g.", r, c, o[r, c], ref[r, c], "diff", o[r, c] - ref[r, c])
        # is the wrong value the product with part of K missing?
        if BIAS:
            bn = b.double().cpu().numpy()
            for c2 in range(max(0, c - 300), min(N, c + 300)):
                if abs(ref[r, c] - bn[c] + bn[c2] - o[r, c]) < 0.02 and not GELU:
                    print(f"    = product + bias[{c2}] (col {c})")
            print("   without bias:", ref[r, c] - bn[c])
        Ad, Wd = A.double().cpu().numpy(), W.double().cpu().numpy()
        for kt in range(K // 32):
            part = Ad[r, kt * 32:(kt + 1) * 32] @ Wd[c, kt * 32:(kt + 1) * 32]
            if abs((ref[r, c] - part) - o[r, c]) < 0.02:
                print(f"    = product without K-tile {kt}")
