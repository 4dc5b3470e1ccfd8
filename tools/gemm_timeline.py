#!/usr/bin/env python3
"""Diagnostic: wall-clock timeline of the workgroups of one pp_gemm launch (-DPP_GEMM_TIMELINE lab build):
when each workgroup entered, ran its K-loop and ended, on which CU, and the idle gaps between consecutive
workgroups of a CU.  usage: gemm_timeline.py --lib lab_tl.so M N K tile [resid] [gelu]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from probpose_pytorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i = sys.argv.index("--lib")
L = C.CDLL(os.path.join(ROOT, "probpose_pytorch_amd", "lib", "exp", sys.argv[i + 1]))
del sys.argv[i:i + 2]
L.pp_gemm.restype = C.c_int
L.pp_gemm.argtypes = [C.POINTER(_lib.GemmArgs), C.c_void_p]
L.pp_last_error.restype = C.c_char_p
M, N, K, tile = (int(v) for v in sys.argv[1:5])
RESID, GELU = "resid" in sys.argv, "gelu" in sys.argv
g = torch.Generator().manual_seed(0)
A = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
W = (torch.randn((N, K), generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
b = torch.randn((N,), generator=g).cuda()
out = torch.randn((M, N), device="cuda") if RESID else torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
NWAVES = 8
stamps = torch.zeros((1 << 14, NWAVES, 8), dtype=torch.int64, device="cuda")
a = _lib.GemmArgs()
a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), out.data_ptr(), b.data_ptr()
a.rowbias = stamps.data_ptr()
a.M, a.N, a.Kd, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
a.batch, a.dtype, a.tile = 1, 1, tile
a.epilogue = (1 | 8 | 16 if RESID else 1) | (2 if GELU else 0) | (1 << 30)
if RESID:
    a.residual = out.data_ptr()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    stamps.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = L.pp_gemm(C.byref(a), st)
    e1.record()
    assert rc == 0, L.pp_last_error()
    torch.cuda.synchronize()
s = stamps.cpu().numpy()
wg = s[:, :, 6].max(axis=1) > 0
s = s[wg]                                   # [workgroups, waves, 8]
live = s[:, :, 6] > 0
big = np.iinfo(np.int64).max
entry = np.where(live, s[:, :, 0], big).min(axis=1)
loop0 = np.where(live, s[:, :, 1], big).min(axis=1)
loop1 = np.where(live, s[:, :, 2], 0).max(axis=1)
end = np.where(live, s[:, :, 3], 0).max(axis=1)
hw = s[:, 0, 4]
xcc = s[:, 0, 5] & 0xF
cu_key = xcc * 65536 + ((hw >> 8) & 0xFFFF & ~0x0)   # cu_id / sh / se bits
t0 = entry.min()
us = lambda ticks: ticks / 100.0                      # 100 MHz -> us
print(f"M={M} N={N} K={K} tile={tile} resid={RESID} gelu={GELU}: {len(s)} workgroups on {len(set(cu_key))} CUs; "
      f"event time {e0.elapsed_time(e1) * 1e3:.1f} us, first entry -> last end {us(end.max() - t0):.1f} us")
print(f"  per workgroup: entry->loop {us((loop0 - entry).mean()):.2f} us, K-loop {us((loop1 - loop0).mean()):.2f} "
      f"(p10 {us(np.percentile(loop1 - loop0, 10)):.2f}, p90 {us(np.percentile(loop1 - loop0, 90)):.2f}), "
      f"epilogue {us((end - loop1).mean()):.2f}, lifetime {us((end - entry).mean()):.2f} us")
gaps, firsts, busy = [], [], []
for key in set(cu_key):
    idx = np.where(cu_key == key)[0]
    idx = idx[np.argsort(entry[idx])]
    firsts.append(us(entry[idx[0]] - t0))
    busy.append(us((end[idx] - entry[idx]).sum()))
    for a_, b_ in zip(idx[:-1], idx[1:]):
        gaps.append(us(entry[b_] - end[a_]))
gaps = np.array(gaps) if gaps else np.zeros(1)
print(f"  first workgroup of a CU starts {np.mean(firsts):.2f} us after the earliest (max {np.max(firsts):.2f}); "
      f"gap between consecutive workgroups on a CU: mean {gaps.mean():.2f} us, p50 {np.median(gaps):.2f}, "
      f"p90 {np.percentile(gaps, 90):.2f}; busy per CU {np.mean(busy):.1f} us (min {np.min(busy):.1f}, max {np.max(busy):.1f})")
rounds = np.bincount(np.unique(cu_key, return_inverse=True)[1])
print(f"  workgroups per CU: min {rounds.min()} max {rounds.max()}; last end per CU spread: "
      f"{us(np.percentile([end[cu_key == k].max() for k in set(cu_key)], 5) - t0):.1f} .. "
      f"{us(max(end[cu_key == k].max() for k in set(cu_key)) - t0):.1f} us")
