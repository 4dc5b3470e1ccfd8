#!/usr/bin/env python3
"""Frame -> keypoints: front end (host plan + crop_resize) + ViT-B forward + decode for 64 person boxes of a
1080p frame, bf16, eager launches (the plan changes per frame, so the front end is not graph-captured; the forward
+ decode is replayed from a HIP graph that reads the crop buffer)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from probpose_pytorch_amd import frontend, ops

ops.AUTOTUNE = True
cfg = dict(bench.CONFIGS["vit_b"])
dev = torch.device("cuda", 0)
model, codec, _ = bench.build(cfg, torch.bfloat16, dev)
rng = np.random.default_rng(0)
H, W, n = 1080, 1920, 64
frame = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
size = (192, 256)


def boxes_for(i):
    r = np.random.default_rng(100 + i)
    return [(float(r.uniform(0, W - 450)), float(r.uniform(0, H - 700)), float(r.uniform(120, 450)),
             float(r.uniform(250, 700))) for _ in range(n)]


crops = torch.empty((n, 3, size[1], size[0]), dtype=torch.float32, device=dev)
with torch.no_grad():
    for _ in range(3):
        codec.decode_device(model(crops))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        codec.decode_device(model(crops))
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        out = codec.decode_device(model(crops))

    def one_frame(i):
        t0 = time.perf_counter()
        plan = frontend.FrontendPlan(frontend.round_boxes(boxes_for(i)), size, dev)     # host tables + upload
        t1 = time.perf_counter()
        crops.copy_(frontend.crop_resize(frame, None, size, plan=plan))
        g.replay()
        return t1 - t0

    for i in range(3):
        one_frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    F = 30
    for i in range(F):
        host += one_frame(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / F
print(f"frame -> keypoints, {n} persons per 1080p frame: {dt * 1e3:.2f} ms per frame = {1 / dt:.0f} frames/s = {n / dt:.0f} persons/s "
      f"(host plan build + upload {host / F * 1e3:.2f} ms of it, not overlapped)")
