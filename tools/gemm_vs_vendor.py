#!/usr/bin/env python3
"""Yardstick (measurement only, never imported by the package): the vendor bf16 GEMM that torch ships
(`torch.mm` / `torch.bmm` -> hipBLASLt / rocBLAS) beside pp_gemm on the GEMM shapes of the ViT-B bs-64 step,
interleaved in ONE process on one device (box-to-box spread is 3-10 %), random operands, medians.

The vendor call is the bare product (no bias / GELU / residual): it is a lower bound on what a library
call for the layer would cost.  pp_gemm is timed twice: bare (same work as the vendor call) and with the
layer's own epilogue.  Output: one row per shape -> profiles/r03_gemm_vs_vendor.txt.
usage: gemm_vs_vendor.py [--rounds 7] [--inner 10]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probpose_pytorch_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--inner", type=int, default=10)
ap.add_argument("--tiles", default="2,3,4,5,6,7,9,10,13,14,18,19,20")
ap.add_argument("--shapes", default="", help="comma-separated shape names (default: all)")
ap.add_argument("--no-vendor", action="store_true")
args = ap.parse_args()
TILES = [int(t) for t in args.tiles.split(",")]

# (name, M, N, K, batch, epilogue of the layer)
R, G, F = _lib.EPI_RESIDUAL | _lib.EPI_OUT_F32 | _lib.EPI_BIAS, _lib.EPI_BIAS | _lib.EPI_GELU, _lib.EPI_BIAS
SHAPES = [
    ("qkv", 12288, 2304, 768, 1, F),
    ("proj", 12288, 768, 768, 1, R),
    ("fc1", 12288, 3072, 768, 1, G),
    ("fc2", 12288, 768, 3072, 1, R),
    ("aux_conv0", 12288, 3072, 6912, 1, F),        # the product's form is an implicit 3x3 convolution (gather); same flops
    ("aux_stage1", 1024, 768, 2304, 12, _lib.EPI_OUT_F32),
    ("aux_stage2", 256, 768, 768, 36, _lib.EPI_OUT_F32),
    ("deconv_parity", 12288, 256, 3072, 4, _lib.EPI_BIAS | _lib.EPI_RELU),
]
# NOT in the list: deconv1 (M 49152, N 256, K 1024, batch 4).  torch.bmm on it ends in "Memory access fault by GPU" inside
# the vendor library on this image (reproduced with no pp_gemm call in the process, gpurun_out/r03a/iso_vendor.err);
# the product runs that layer through pp_gemm only.

L = _lib.lib()
st = _lib.stream_ptr()
g = torch.Generator().manual_seed(0)


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, inner):
    fn()
    s, e = ev(), ev()
    s.record()
    for _ in range(inner):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / inner      # us


print(f"# device: {torch.cuda.get_device_name(0)}; torch {torch.__version__}; rounds {args.rounds} x inner {args.inner}, medians; "
      f"random N(0,1) activations, N(0,1/K) weights; TF = 2*M*N*K*batch / time")
print(f"# {'shape':16s} {'M':>6s} {'N':>5s} {'K':>5s} {'b':>3s} | {'vendor us':>9s} {'TF':>6s} | {'pp bare us':>10s} {'TF':>6s} {'tile':>4s} | "
      f"{'pp layer us':>11s} {'TF':>6s} {'tile':>4s} | pp bare / vendor")
want = set(args.shapes.split(",")) if args.shapes else {s[0] for s in SHAPES}
for name, M, N, K, b, epi in SHAPES:
    if name not in want:
        continue
    A = torch.randn((b, M, K), generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn((b, N, K), generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    Wt = W.transpose(1, 2)                       # [b, K, N] view, K-contiguous rows of W: the "NT" library form
    bias = torch.randn((b, N), generator=g).cuda()
    out16 = torch.empty((b, M, N), dtype=torch.bfloat16, device="cuda")
    out32 = torch.randn((b, M, N), device="cuda") if (epi & _lib.EPI_OUT_F32) else None
    vout = torch.empty((b, M, N), dtype=torch.bfloat16, device="cuda")

    def vendor():
        if args.no_vendor:
            return
        if b == 1:
            torch.mm(A[0], Wt[0], out=vout[0])
        else:
            torch.bmm(A, Wt, out=vout)

    def mk(epilogue, tile):
        a = _lib.GemmArgs()
        o = out32 if (epilogue & _lib.EPI_OUT_F32) else out16
        a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), o.data_ptr(), bias.data_ptr()
        if epilogue & _lib.EPI_RESIDUAL:
            a.residual = o.data_ptr()
        a.M, a.N, a.Kd, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
        a.batch, a.strideA, a.strideW, a.strideC, a.strideBias = b, M * K, N * K, M * N, N
        a.dtype, a.tile, a.epilogue = 1, tile, epilogue
        return a

    cands = {}
    for mode, epilogue in (("bare", 0), ("layer", epi)):
        for t in TILES:
            a = mk(epilogue, t)
            if L.pp_gemm(C.byref(a), st) == 0:
                cands[(mode, t)] = a
                torch.cuda.synchronize()          # a faulting configuration is named by the last line on stderr
                print(f"ok {name} {mode} tile {t}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    tv, tp = [], {k: [] for k in cands}
    for _ in range(args.rounds):
        tv.append(timed(vendor, args.inner))
        for k, a in cands.items():
            tp[k].append(timed(lambda: L.pp_gemm(C.byref(a), st), args.inner))
    med = lambda v: sorted(v)[len(v) // 2]
    flop = 2.0 * M * N * K * b
    v_us = med(tv)
    best = {}
    for mode in ("bare", "layer"):
        ks = [k for k in cands if k[0] == mode]
        if not ks:
            best[mode] = (float("nan"), -1)
            continue
        kb = min(ks, key=lambda k: med(tp[k]))
        best[mode] = (med(tp[kb]), kb[1])
    # correctness of the yardstick itself: same product
    err = float("nan")
    if best["bare"][1] >= 0 and not args.no_vendor:
        a = mk(0, best["bare"][1])
        L.pp_gemm(C.byref(a), st)
        vendor()
        torch.cuda.synchronize()
        err = (out16.float() - vout.float()).abs().max().item()
    print(f"  {name:16s} {M:6d} {N:5d} {K:5d} {b:3d} | {v_us:9.1f} {flop / v_us / 1e6:6.0f} | {best['bare'][0]:10.1f} "
          f"{flop / best['bare'][0] / 1e6:6.0f} {best['bare'][1]:4d} | {best['layer'][0]:11.1f} {flop / best['layer'][0] / 1e6:6.0f} "
          f"{best['layer'][1]:4d} | {best['bare'][0] / v_us:5.2f}   (max |pp - vendor| = {err:.3g})")
    print("      all pp tiles, bare: " + "  ".join(f"t{k[1]}={med(tp[k]):.1f}" for k in cands if k[0] == "bare"))
    del A, W, out16, out32, vout
    torch.cuda.empty_cache()
