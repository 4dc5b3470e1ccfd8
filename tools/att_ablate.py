#!/usr/bin/env python3
"""Diagnostic: one-shot attention kernel with one ingredient removed (tools/att_ablate.sh builds the variants):
0 = complete, 1 = no K / V loads, 2 = no exponentials, 3 = loads and stores only."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, N, heads, hd = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 192, 12, 64)
Cc = heads * hd
qkv = torch.randn((B * N, 3 * Cc), device="cuda").to(torch.bfloat16)
out = torch.zeros((B * N, Cc), dtype=torch.bfloat16, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for a, name in enumerate(["complete", "no K/V loads", "no exponentials", "loads + stores only", "scalar fma/add softmax"]):
    L = C.CDLL(os.path.join(ROOT, "probpose_pytorch_amd", "lib", "diag", f"libpp_att_abl{a}.so"))
    L.pp_attention.restype = C.c_int
    L.pp_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for _ in range(5):
        assert L.pp_attention(qkv.data_ptr(), out.data_ptr(), B, N, heads, hd, 1, st) == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        L.pp_attention(qkv.data_ptr(), out.data_ptr(), B, N, heads, hd, 1, st)
    e.record()
    torch.cuda.synchronize()
    print(f"{name:22s} {s.elapsed_time(e) / 50 * 1e3:7.1f} us", flush=True)
