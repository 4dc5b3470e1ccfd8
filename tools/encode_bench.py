#!/usr/bin/env python3
"""Time ProbMap.encode_device (pp_encode_probmaps) for a batch of crops: a write-bound kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

g.build()
from probpose_pytorch_amd import _lib, ProbMap

for B, K, in_size, hm in ((64, 17, (192, 256), (48, 64)), (1024, 17, (192, 256), (48, 64)), (128, 133, (288, 384), (72, 96))):
    rng = np.random.default_rng(0)
    sig = rng.uniform(0.02, 0.11, K)
    pm = ProbMap(in_size, hm, sig, sigma=None)
    kp = np.stack([rng.uniform(0, in_size[0], (B, K)), rng.uniform(0, in_size[1], (B, K))], -1).astype(np.float32)
    d_kp = torch.from_numpy((kp / pm.scale_factor).astype(np.float32)).cuda()
    d_vis = torch.ones((B, K), device="cuda")
    d_s = torch.from_numpy(pm._two_s()).cuda()
    W, H = hm
    heat = torch.empty((B, K, H, W), device="cuda")
    wts = torch.empty((B, K), device="cuda")
    L = _lib.lib()
    run = lambda: L.pp_encode_probmaps(_lib.ptr(d_kp), _lib.ptr(d_vis), _lib.ptr(d_s), B, K, H, W, _lib.ptr(heat),
                                       _lib.ptr(wts), _lib.stream_ptr())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        run()
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    nbytes = heat.numel() * 4
    print(f"B={B} K={K} {H}x{W}: {t * 1e6:.1f} us, {nbytes / 1e6:.1f} MB written = {nbytes / t / 1e9:.0f} GB/s, {B / t:.0f} crops/s")
