#!/usr/bin/env python3
"""Diagnostic: phase cycle shares of the MFMA attention kernel (-DPP_ATT_STAMPS build)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "probpose_pytorch_amd", "lib", "diag", "libpp_att_stamps.so"))
L.pp_attention.restype = C.c_int
L.pp_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
B, N, heads, hd = 64, 192, 12, 64
Cc = heads * hd
qkv = torch.randn((B * N, 3 * Cc), device="cuda").to(torch.bfloat16)
nblk = B * heads
out = torch.zeros((B * N * Cc + nblk * 8 * 4 + 64,), dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    rc = L.pp_attention(qkv.data_ptr(), out.data_ptr(), B, N, heads, hd, 1, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
torch.cuda.synchronize()
st = out[B * N * Cc: B * N * Cc + nblk * 32].view(torch.int64).cpu().numpy().reshape(nblk, 8)
names = ["stage K/V + Q loads", "barrier", "S, softmax, PV", "normalise + store", "total"]
for i, n in enumerate(names):
    print(f"  {n:22s} mean {st[:, i].mean():9.0f} cycles ({100 * st[:, i].mean() / st[:, 4].mean():5.1f} %)")
print("  span", st[:, 5].max() + st[st[:, 5].argmax(), 4] - st[:, 5].min(), "cycles")
