#!/usr/bin/env python3
"""Diagnostic: time the decode alone on the heatmaps the bench model (random weights: clamped plateaus, flat maps)
actually produces, and on peaked synthetic maps of the same shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "vit_b"]
dev = torch.device("cuda", 0)
model, codec, sd = bench.build(cfg, torch.bfloat16, dev)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, W = cfg.get("img", (256, 192))
x = torch.rand((B, 3, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
with torch.no_grad():
    hm = model(x)[0].float().contiguous()


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


from probpose_pytorch_amd import _lib, heatmap as hmod

res = {}
for rnd in range(5):                 # interleaved rounds, medians
    for name, flags in (("default", 0), ("wave-per-map + list kernel", _lib.DECODE_WAVE),
                        ("all-pixel float64", _lib.DECODE_ALL_PIXEL)):
        hmod.DECODE_FLAGS = flags
        res.setdefault(name, []).append(t(lambda: codec.probmap.decode_device(hm)))
hmod.DECODE_FLAGS = 0
for name, v in res.items():
    print(f"{name:38s} model heatmaps {tuple(hm.shape)}: {sorted(v)[len(v) // 2]:8.1f} us (Python call + launch included)", flush=True)

