#!/usr/bin/env python3
"""Time the fused decode kernel vs batch size and kernel radius (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from probpose_pytorch_amd import _lib

if "--lib" in sys.argv:      # A/B an experimental build of the library (tools/build_exp.sh)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from probpose_pytorch_amd.heatmap import decode_on_device, oks_tap_table

COCO = np.array([.026, .025, .025, .035, .035, .079, .079, .072, .072, .062, .062, .107, .107, .087, .087, .089, .089])


def bench(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for (K, H, W, sig, tag) in [(17, 64, 48, COCO, "coco"), (17, 64, 48, np.full(17, 0.01), "r=2"),
                            (17, 64, 48, np.full(17, 0.5), "r=9"), (133, 96, 72, np.random.default_rng(133).uniform(0.02, 0.11, 133), "k133")]:
    print(tag, "radii", sorted(set(oks_tap_table(K, H, W, sig)[1].tolist())))
    for B in (1, 16, 64, 256, 1024):
        if K == 133 and B > 128:
            continue
        hm = torch.rand((B, K, H, W), device="cuda")
        t = bench(lambda: decode_on_device(hm, sig, den=(W - 1, H - 1), input_size=(W * 4, H * 4)))
        by = B * K * H * W * 4
        print(f"  B={B:5d}  {t:8.1f} us   {by / t / 1e3:8.1f} GB/s   {B * K / t:6.2f} maps/us")
