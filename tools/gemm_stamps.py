#!/usr/bin/env python3
"""Diagnostic: cycle shares of the GEMM K-loop phases from the -DPP_GEMM_STAMPS build
(probpose_pytorch_amd/lib/diag/libpp_gemm_stamps.so, built by tools/build_diag.sh).
args: M N K tile"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from probpose_pytorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBP = os.path.join(ROOT, "probpose_pytorch_amd", "lib", "diag", "libpp_gemm_stamps.so")
if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    LIBP = os.path.join(ROOT, "probpose_pytorch_amd", "lib", "exp", sys.argv[i + 1])
    del sys.argv[i:i + 2]
L = C.CDLL(LIBP)
L.pp_gemm.restype = C.c_int
L.pp_gemm.argtypes = [C.POINTER(_lib.GemmArgs), C.c_void_p]
L.pp_last_error.restype = C.c_char_p

M, N, K, tile = (int(v) for v in sys.argv[1:5])
RESID = len(sys.argv) > 5 and sys.argv[5] == "resid"
g = torch.Generator().manual_seed(0)
A = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
W = (torch.randn((N, K), generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
b = torch.randn((N,), generator=g).cuda()
out = torch.randn((M, N), device="cuda") if RESID else torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
stamps = torch.zeros((1 << 16, 8, 8), dtype=torch.int64, device="cuda")
a = _lib.GemmArgs()
a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), out.data_ptr(), b.data_ptr()
a.rowbias = stamps.data_ptr()
a.M, a.N, a.Kd, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
a.batch, a.dtype, a.epilogue, a.tile = 1, 1, (1 | 8 | 16 if RESID else 1) | (1 << 30) | (2 if 'gelu' in sys.argv else 0), tile
if RESID:
    a.residual = out.data_ptr()
for _ in range(3):
    rc = L.pp_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.pp_last_error()
torch.cuda.synchronize()
s_all = stamps.cpu().numpy().reshape(-1, 8, 8)          # [workgroup, wave, field]
names = ["prologue", "vmcnt wait", "barrier", "stage issue", "compute", "epilogue", "total"]
if tile == 13:   # persistent: per-workgroup totals over all its tiles
    names = ["stream advance", "vmcnt wait", "barrier", "deferred slice", "compute", "last epilogue", "total"]
if tile == 10:   # ping-pong: fields are frag-read issue / wait+barrier / DMA issue / MFMA, per wave role
    names = ["prologue", "frag reads", "wait+barrier", "DMA issue", "MFMA", "epilogue", "total"]
print(f"M={M} N={N} K={K} tile={tile}: K-tiles = {K // 64}")
for role, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    s = s_all[:, sl].reshape(-1, 8)
    s = s[s[:, 6] > 0]
    if not len(s):
        continue
    print(f" {role}: {len(s)} waves stamped")
    for i, n in enumerate(names):
        print(f"  {n:12s} mean {s[:, i].mean():10.0f} cycles  ({100 * s[:, i].mean() / s[:, 6].mean():5.1f} %)   p50 {np.median(s[:, i]):9.0f}")
s = s_all.reshape(-1, 8)
s = s[s[:, 6] > 0]
span = s[:, 7].max() + s[s[:, 7].argmax(), 6] - s[:, 7].min()
print(f"  kernel span {span} cycles (s_memtime)")
