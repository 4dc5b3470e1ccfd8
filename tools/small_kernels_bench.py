#!/usr/bin/env python3
"""Time (and, under `rocprofv3 --pmc`, count the HBM bytes of) the memory-bound kernels beside the step's big three:
sparsemax_rows_kernel, dark_decode_kernel, heatmap_argmax_kernel, pck_counts_kernel (Python call overhead included in the
event times below: the kernel-trace of the same run gives the kernels' own durations).
HIP events around N back-to-back launches on resident buffers; GB/s against the ALGORITHMIC bytes (each map read once,
written once where the kernel writes maps).  usage: small_kernels_bench.py [--iters 50] [--once]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import probpose_oracle as orc
from probpose_pytorch_amd import ArgMaxProbMap, _lib, metrics, ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--once", action="store_true", help="one launch of each kernel (for rocprofv3 --pmc passes)")
args = ap.parse_args()
L = _lib.lib()


def timed(fn):
    if args.once:
        fn()
        torch.cuda.synchronize()
        return float("nan")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / args.iters * 1e3


def row(name, shape, us, nbytes):
    print(f"{name:28s} {shape:22s} {us:9.1f} us  {nbytes / 1e6:9.2f} MB algorithmic  {nbytes / us / 1e3 if us == us else float('nan'):8.1f} GB/s  "
          f"({nbytes / us / 1e3 / 8000 if us == us else float('nan'):.3f} of 8 TB/s)", flush=True)


for (B, K, H, W) in ((64, 17, 64, 48), (1024, 17, 64, 48), (128, 133, 96, 72)):
    hm = torch.from_numpy(orc.synthetic_heatmaps(min(B, 32), K, H, W, 4321, "peaked")).cuda()
    hm = hm.repeat((B + hm.shape[0] - 1) // hm.shape[0], 1, 1, 1)[:B].contiguous()
    shape = f"{B}x{K}x{H}x{W}"
    maps_bytes = float(B * K * H * W * 4)
    # Sparsemax over H*W (+ * normalize, clamp), in place: read + write every pixel
    logits = (hm * 3 - 1).reshape(B * K, H * W).contiguous()
    work = logits.clone()
    row("sparsemax_rows_kernel", shape, timed(lambda: ops.sparsemax_rows(work, 1.0)), 2 * maps_bytes)   # (idempotent on its output)
    # ArgMax + DARK-UDP decode: the map read once
    codec = ArgMaxProbMap((4 * W, 4 * H), (W, H), blur_kernel_size=11)
    row("dark_decode_kernel", shape, timed(lambda: codec.decode_device(hm)), maps_bytes)
    # plain arg-max (get_heatmap_maximum)
    locs = torch.empty((B, K, 2), device="cuda")
    vals = torch.empty((B, K), device="cuda")
    st = _lib.stream_ptr()
    row("heatmap_argmax_kernel", shape, timed(lambda: L.pp_heatmap_argmax(_lib.ptr(hm), B * K, H, W, _lib.ptr(locs), _lib.ptr(vals), st)),
        maps_bytes)
# PCK counts over many pairs: 20 B per pair
for (N, K) in ((4096, 17), (262144, 17), (65536, 133)):
    p = torch.rand((N, K, 2), device="cuda") * 48
    g = p + torch.randn((N, K, 2), device="cuda")
    m = torch.rand((N, K), device="cuda") > 0.2
    nrm = np.tile(np.array([[64.0, 48.0]]), (N, 1))
    row("pck_counts_kernel (+ host)", f"{N}x{K} pairs", timed(lambda: metrics.pck_counts(p, g, m, 0.05, nrm)), float(N * K * 17))
