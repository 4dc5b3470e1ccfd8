#!/usr/bin/env python3
"""A/B of pp_attention between the shipped library and lib/exp/<name>.so (bf16, the one-shot kernel's shapes), interleaved
in one process, medians.  usage: att_ab.py att_before.so"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probpose_pytorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"shipped": C.CDLL(_lib.LIB_PATH), sys.argv[1]: C.CDLL(os.path.join(ROOT, "probpose_pytorch_amd", "lib", "exp", sys.argv[1]))}
for L in libs.values():
    L.pp_attention.restype = C.c_int
    L.pp_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, B, N, heads, hd in [("vit_b bs64", 64, 192, 12, 64), ("vit_l bs256", 256, 192, 16, 64), ("vit_s bs64 (hd 32)", 64, 192, 12, 32),
                              ("vit_h 384x288 bs128 (hd 80, streaming)", 128, 432, 16, 80), ("vit_l 384x288 bs64 (hd 64, streaming)", 64, 432, 16, 64),
                              ("ragged N = 433, hd 80", 8, 433, 16, 80), ("ragged N = 200, hd 32", 2, 200, 2, 32), ("ragged N = 50, hd 64", 1, 50, 4, 64)]:
    Cc = heads * hd
    qkv = torch.randn((B * N, 3 * Cc), device="cuda").to(torch.bfloat16)
    outs = {k: torch.empty((B * N, Cc), dtype=torch.bfloat16, device="cuda") for k in libs}
    times = {k: [] for k in libs}
    for rnd in range(9):
        for k, L in libs.items():
            call = lambda: L.pp_attention(qkv.data_ptr(), outs[k].data_ptr(), B, N, heads, hd, 1, st)
            assert call() == 0
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                call()
            e.record()
            e.synchronize()
            times[k].append(s.elapsed_time(e) / 20 * 1e3)
    same = torch.equal(*outs.values())
    if not same:
        a, b_ = list(outs.values())
        bad = (a != b_).nonzero()
        print("   differing elements:", bad.shape[0], "first", bad[:6].tolist(), "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:24])
    print(name, {k: round(sorted(v)[len(v) // 2], 2) for k, v in times.items()}, "us; outputs identical:", same)
