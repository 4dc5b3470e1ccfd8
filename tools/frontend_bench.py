#!/usr/bin/env python3
"""Time the front end (host plan build, plan upload, crop_resize_kernel) on a 1080p frame with 64 person boxes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as g

g.build()
from probpose_pytorch_amd import _lib, frontend

if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])

rng = np.random.default_rng(0)
H, W, n = 1080, 1920, 64
img = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
boxes = [(float(rng.uniform(0, W - 450)), float(rng.uniform(0, H - 700)), float(rng.uniform(120, 450)),
          float(rng.uniform(250, 700))) for _ in range(n)]
size = (192, 256)
t0 = time.perf_counter()
for _ in range(5):
    xyxy = frontend.round_boxes(boxes)
    plan = frontend.FrontendPlan(xyxy, size, None)
t_plan = (time.perf_counter() - t0) / 5
plan = frontend.FrontendPlan(frontend.round_boxes(boxes), size, img.device)
out = frontend.crop_resize(img, boxes, size, plan=plan)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    frontend.crop_resize(img, boxes, size, plan=plan)
e.record()
torch.cuda.synchronize()
t_k = s.elapsed_time(e) / 20 * 1e-3
src_bytes = sum(int(round(b[2])) * int(round(b[3])) * 3 for b in boxes)
out_bytes = n * 3 * size[0] * size[1] * 4
print(f"{n} boxes of a {W}x{H} frame -> {size[0]}x{size[1]}: host plan build {t_plan * 1e3:.2f} ms "
      f"({plan.host.nbytes / 1e6:.2f} MB of tables, {plan.n_blocks} workgroups, {plan.lds_bytes} B LDS), "
      f"kernel {t_k * 1e6:.1f} us = {n / t_k:.0f} crops/s, "
      f"{(src_bytes + out_bytes) / t_k / 1e9:.1f} GB/s algorithmic (crop pixels read once + f32 crops written)")
