#!/usr/bin/env python3
"""Diagnostic: how many float32-screen candidates do the bench model's (random-weight) heatmaps have per map?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from oracle import probpose_oracle as orc

cfg = bench.CONFIGS["vit_b"]
dev = torch.device("cuda", 0)
model, codec, sd = bench.build(cfg, torch.bfloat16, dev)
x = torch.rand((8, 3, 256, 192), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
with torch.no_grad():
    hm = model(x)[0].float().cpu().numpy()
B, K, H, W = hm.shape
sig = bench.sigmas_for(K)
n_c, n_eq, sat = [], [], []
for b in range(B):
    _, _, conv = orc.heatmap_expected_value(hm[b], sig, "scipy", return_heatmap=True)
    for k in range(K):
        A = np.abs(hm[b, k]).max()
        c = conv[k]
        n_c.append(int((c >= c.max() - 128 * 2.0 ** -24 * A).sum()))
        n_eq.append(int((c == c.max()).sum()))
        sat.append(float((hm[b, k] >= 1.0).mean()))
n_c, n_eq, sat = np.array(n_c), np.array(n_eq), np.array(sat)
print("maps", len(n_c), "candidates: median", np.median(n_c), "p90", np.percentile(n_c, 90), "max", n_c.max(),
      "frac > 64:", (n_c > 64).mean(), "frac > 8:", (n_c > 8).mean())
print("exact ties at max: median", np.median(n_eq), "max", n_eq.max(), " saturated-pixel share: mean", sat.mean(), "max", sat.max())
print("raw min/max", hm.min(), hm.max(), "zeros share", (hm == 0).mean())
