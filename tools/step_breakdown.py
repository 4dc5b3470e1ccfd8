#!/usr/bin/env python3
"""Per-launch timing of one eager forward+decode step (HIP events around every hooked launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from probpose_pytorch_amd import engine, ops

engine.SERIALIZE_HEAD = True
ops.AUTOTUNE = True
from probpose_pytorch_amd.synthetic import synthetic_crops

cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "vit_b"])
dev = torch.device("cuda", 0)
model, codec, _ = bench.build(cfg, torch.bfloat16, dev)
x = synthetic_crops(cfg["batch"], *cfg["img"], seed=1234).to(dev)
with torch.no_grad():
    for _ in range(3):
        codec.decode_device(model(x))
    torch.cuda.synchronize()
    runs = []
    for _ in range(5):
        prof = []
        ops.set_profile(prof)
        codec.decode_device(model(x))
        torch.cuda.synchronize()
        ops.set_profile(None)
        runs.append([(n, w, s.elapsed_time(e) * 1e3, i) for n, w, s, e, i in prof])
best = runs[-1]
seen = {}
tot = 0.0
for k, (n, w, t, info) in enumerate(best):
    t = min(r[k][2] for r in runs)
    tot += t
    key = (n, info)
    a = seen.setdefault(key, [0, 0.0, w])
    a[0] += 1
    a[1] += t
print(f"sum of hooked launches: {tot / 1e3:.3f} ms")
for (n, info), (c, t, w) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
    rate = w / (t / c) / 1e6 if n in ("gemm", "attention") else w / (t / c) / 1e3
    unit = "TFLOP/s" if n in ("gemm", "attention") else "GB/s"
    print(f"{n:10s} x{c:3d}  {t / c:8.1f} us each  {t:9.1f} us total  {rate:8.1f} {unit}  {info}")
