#!/usr/bin/env python3
"""fp8 vs bf16 heatmap deviation of the synthetic ViT-B model, with and without the fp8 proj GEMM."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from probpose_pytorch_amd import engine
from probpose_pytorch_amd.synthetic import synthetic_crops

cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "vit_b"])
dev = torch.device("cuda", 0)
x = synthetic_crops(16, *cfg["img"], seed=7).to(dev)
with torch.no_grad():
    model, codec, _ = bench.build(cfg, torch.float32, dev)
    ref32 = model(x)[0].float().clone()
    model.set_compute_dtype(torch.bfloat16)
    ref = model(x)[0].float().clone()
    k16 = codec.decode_device(model(x))["kpts"].clone()
    print(f"bf16 vs fp32: mean |d| {float((ref - ref32).abs().mean()):.4f} max {float((ref - ref32).abs().max()):.3f}")
    for flag in (False, True):
        engine.FP8_PROJ = flag
        model, codec, _ = bench.build(cfg, torch.float8_e4m3fn, dev)
        out = model(x)
        d = (out[0].float() - ref32).abs()
        k8 = codec.decode_device(out)["kpts"]
        kd = (k8 - k16).abs().amax(-1)
        print(f"fp8 (proj fp8 = {flag}) vs fp32: mean |d| {float(d.mean()):.4f} max {float(d.max()):.3f}; keypoints vs bf16: "
              f"median {float(kd.median()):.3f} px, {float((kd < 1.0).double().mean()) * 100:.1f} % within 1 px")
