#!/usr/bin/env python3
"""Run one pp_gemm shape repeatedly (for rocprofv3 --pmc passes).  args: M N K tile iters [resid]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probpose_pytorch_amd import ops

M, N, K, tile, iters = (int(v) for v in sys.argv[1:6])
g = torch.Generator().manual_seed(0)
A = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
W = (torch.randn((N, K), generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
b = torch.randn((N,), generator=g).cuda()
RESID = len(sys.argv) > 6 and sys.argv[6] == "resid"
out = torch.randn((M, N), device="cuda") if RESID else torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
for _ in range(iters):
    ops.linear(A, W, b, out=out, tile=tile, residual=out if RESID else None)
torch.cuda.synchronize()
print("done")
