#!/usr/bin/env python3
"""A/B experimental GEMM builds (tools/build_lab.sh) on the ViT-B bs64 shapes: interleaved rounds in one
process, correctness of every (lib, tile) against a float32 torch matmul of the same bf16 operands.
usage: gemm_lab.py [--tiles 3,10] [--stamps] lib1.so [lib2.so ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from probpose_pytorch_amd import _lib

args = sys.argv[1:]
tiles = (3, 10)
if "--tiles" in args:
    i = args.index("--tiles")
    tiles = tuple(int(t) for t in args[i + 1].split(","))
    del args[i:i + 2]
stamps_mode = "--stamps" in args
if stamps_mode:
    args.remove("--stamps")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = []
for path in args:
    if not os.path.isabs(path):
        path = os.path.join(ROOT, "probpose_pytorch_amd", "lib", "exp", path)
    L = C.CDLL(path)
    L.pp_gemm.restype = C.c_int
    L.pp_gemm.argtypes = [C.POINTER(_lib.GemmArgs), C.c_void_p]
    L.pp_last_error.restype = C.c_char_p
    libs.append((os.path.basename(path), L))

M = 12288
SHAPES = [("qkv", 2304, 768, 0, False), ("proj", 768, 768, 0, True), ("fc1+gelu", 3072, 768, 2, False),
          ("fc2", 768, 3072, 0, True)]
g = torch.Generator().manual_seed(0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make(N, K, epi, resid):
    A = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn((N, K), generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    b = torch.randn((N,), generator=g).cuda()
    res0 = torch.randn((M, N), generator=g).cuda() if resid else None
    out = torch.empty((M, N), dtype=torch.float32 if resid else torch.bfloat16, device="cuda")
    a = _lib.GemmArgs()
    a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), out.data_ptr(), b.data_ptr()
    a.M, a.N, a.Kd, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
    a.batch, a.dtype = 1, 1
    a.epilogue = 1 | epi | ((8 | 16) if resid else 0)
    if resid:
        a.residual = out.data_ptr()
    return A, W, b, res0, out, a


def reference(A, W, b, res0, epi):
    ref = A.float() @ W.float().t() + b
    if epi & 2:
        ref = torch.nn.functional.gelu(ref)
    if res0 is not None:
        ref = ref + res0
    return ref


for name, N, K, epi, resid in SHAPES:
    A, W, b, res0, out, a = make(N, K, epi, resid)
    ref = reference(A, W, b, res0, epi)
    rows = {}
    for lname, L in libs:
        for tile in tiles:
            a.tile = tile
            if resid:
                out.copy_(res0)
            rc = L.pp_gemm(C.byref(a), stream)
            if rc != 0:
                rows[(lname, tile)] = None
                continue
            torch.cuda.synchronize()
            err = float((out.float() - ref).abs().max())
            tol = 2e-3 if resid else 6e-2
            rows[(lname, tile)] = [err, err < tol, []]
    for rnd in range(5):
        for (lname, tile), r in rows.items():
            if r is None:
                continue
            L = dict(libs)[lname]
            a.tile = tile
            for _ in range(2):
                L.pp_gemm(C.byref(a), stream)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                L.pp_gemm(C.byref(a), stream)
            e.record()
            e.synchronize()
            r[2].append(s.elapsed_time(e) / 20 * 1e3)
    for (lname, tile), r in rows.items():
        if r is None:
            print(f"{name:9s} {lname:24s} tile {tile:2d}: not applicable")
            continue
        t = np.array(r[2])
        fl = 2.0 * M * N * K
        print(f"{name:9s} {lname:24s} tile {tile:2d}: median {np.median(t):7.1f} us (min {t.min():7.1f})  "
              f"{fl / np.median(t) / 1e6:7.1f} TF   max|err| {r[0]:.2e} {'ok' if r[1] else 'WRONG'}")
