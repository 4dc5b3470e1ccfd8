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
LDA0, LDW0 = "lda0" in sys.argv, "ldw0" in sys.argv      # every A / W row aliases row 0: operand traffic served by L1 / L2 (what-if)
# what limits the clock?  "batch8": the problem as 8 batch entries of M/8 rows (control: distinct data per entry);
# "l2res": with batch8, every entry reads the SAME A / W (stride 0): random data, but the operands stay in every XCD's L2;
# "samerows": normal addressing, every row of A (and of W) holds the same values: same traffic, low-entropy MFMA operands
BATCH8, L2RES, SAMEROWS = "batch8" in sys.argv, "l2res" in sys.argv, "samerows" in sys.argv
g = torch.Generator().manual_seed(0)
A = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
W = (torch.randn((N, K), generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
if SAMEROWS:
    A = A[:1].expand(M, K).contiguous()
    W = W[:1].expand(N, K).contiguous()
b = torch.randn((N,), generator=g).cuda()
out = torch.randn((M, N), device="cuda") if RESID else torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
NWAVES = 8
stamps = torch.zeros((1 << 14, NWAVES, 8), dtype=torch.int64, device="cuda")
a = _lib.GemmArgs()
a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), out.data_ptr(), b.data_ptr()
a.rowbias = stamps.data_ptr()
a.M, a.N, a.Kd, a.lda, a.ldw, a.ldc = M, N, K, 0 if LDA0 else K, 0 if LDW0 else K, N
a.batch, a.dtype, a.tile = 1, 1, tile
if BATCH8:
    a.M, a.batch = M // 8, 8
    a.strideA, a.strideW, a.strideC = (0 if L2RES else (M // 8) * K), 0, (M // 8) * N
a.epilogue = (1 | 8 | 16 if RESID else 1) | (2 if GELU else 0) | (1 << 30)
if RESID:
    a.residual = out.data_ptr()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(200):          # hold the device under load first: the clock settles after ~0.1 s
    L.pp_gemm(C.byref(a), st)
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
cyc = np.where(live, s[:, :, 6] - 1, 0).max(axis=1)
wave_ticks = np.where(live, s[:, :, 2] - s[:, :, 1], 1).max(axis=1)
clk_ghz = cyc / (wave_ticks / 100.0) / 1e3          # shader cycles per us of the 100 MHz wall clock
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
print(f"  in-kernel clock over the K-loop: median {np.median(clk_ghz):.3f} GHz (p10 {np.percentile(clk_ghz, 10):.3f}, "
      f"p90 {np.percentile(clk_ghz, 90):.3f}); K-loop {np.median(cyc) / (K // 64):.0f} cycles per 64-deep K-tile"
      f"{' [lda0]' if LDA0 else ''}{' [ldw0]' if LDW0 else ''}{' [batch8]' if BATCH8 else ''}{' [l2res]' if L2RES else ''}"
      f"{' [samerows]' if SAMEROWS else ''}")
if tile in (13, 18, 19):
    epi = np.where(live, s[:, :, 7], 0).max(axis=1)
    print(f"  persistent stream: {np.median(cyc):.0f} cycles per workgroup, of which epilogues {np.median(epi):.0f} "
          f"({np.median(epi) / np.median(cyc):.1%}); tiles per workgroup ~{M * N / 192 / 192 / len(s):.2f}")
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
