#!/usr/bin/env python3
"""Time pp_gemm on the ViT-B bs64 shapes (and a few epilogue variants) with HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as g

g.build()
from probpose_pytorch_amd import _lib, ops

if "--lib" in sys.argv:      # A/B an experimental build of the library (tools/build_exp.sh)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
TILES = tuple(int(t) for t in sys.argv[sys.argv.index("--tiles") + 1].split(",")) if "--tiles" in sys.argv else (3,)


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


def main():
    dt = torch.bfloat16
    M = 12288
    gen = torch.Generator().manual_seed(0)
    rows = []
    for name, N, K, epi, resid in [("qkv", 2304, 768, 0, False), ("proj", 768, 768, 0, True),
                                   ("fc1+gelu", 3072, 768, ops.EPI_GELU, False), ("fc1 nogelu", 3072, 768, 0, False),
                                   ("fc1 relu", 3072, 768, ops.EPI_RELU, False),
                                   ("fc2", 768, 3072, 0, True), ("fc2 noresid", 768, 3072, 0, False),
                                   ("deconv0-like N256 K3072", 256, 3072, ops.EPI_RELU, False),
                                   ("deconv1-like M49152 N256 K1024", 256, 1024, ops.EPI_RELU, False),
                                   ("big 4096x4096x4096", 4096, 4096, 0, False)]:
        Mm = 4096 if name.startswith("big") else (49152 if "M49152" in name else M)
        A = torch.randn((Mm, K), generator=gen).to(dt).cuda()
        W = (torch.randn((N, K), generator=gen) * K ** -0.5).to(dt).cuda()
        b = torch.randn((N,), generator=gen).cuda()
        res = torch.randn((Mm, N), generator=gen).cuda() if resid else None
        out = res if resid else torch.empty((Mm, N), dtype=dt, device="cuda")
        for tile in TILES:
            t = bench(lambda: ops.linear(A, W, b, out=out, epilogue=epi, residual=res, tile=tile))
            rows.append((name, tile, t, 2.0 * Mm * N * K / t / 1e6))
    for r in rows:
        print(f"{r[0]:32s} tile={r[1]}  {r[2]:8.1f} us  {r[3]:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
