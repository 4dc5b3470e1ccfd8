#!/usr/bin/env python3
"""Experiment: ViT-B bs64 step as one kernel chain vs two half-batch chains on two streams
(engine.DUAL_CHAIN), with the half-batch GEMM tiles (a) autotuned standalone, (b) copied from the
full-batch winners."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from probpose_pytorch_amd import engine, ops
from probpose_pytorch_amd.synthetic import synthetic_crops

ops.AUTOTUNE = True
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "vit_b"])
dev = torch.device("cuda", 0)
model, codec, _ = bench.build(cfg, torch.bfloat16, dev)
B = cfg["batch"]
x = synthetic_crops(B, *cfg["img"]).to(dev)


def step():
    return codec.decode_device(model(x))


def timeit(iters=20):
    with torch.no_grad():
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = step()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters, out


t1, o1 = timeit()
print(f"single chain              : {t1:.3f} ms", flush=True)
flat = lambda o: [t for t in (o.values() if isinstance(o, dict) else o) if torch.is_tensor(t)]
ref = [t.clone() for t in flat(o1)]
full = dict(ops._TUNE_CACHE)
engine.DUAL_CHAIN = True
t2, o2 = timeit()
same = all(torch.equal(a, b) for a, b in zip(ref, flat(o2)))
print(f"dual chain, tiles tuned   : {t2:.3f} ms   outputs identical: {same}", flush=True)
print("  half-batch tiles:", {k[:3]: v for k, v in ops._TUNE_CACHE.items() if k not in full}, flush=True)
for k, v in full.items():
    if k[0] == B * 192 or k[0] == B * (cfg["img"][0] // 16) * (cfg["img"][1] // 16):
        ops._TUNE_CACHE[(k[0] // 2,) + k[1:]] = v
t3, o3 = timeit()
same = all(torch.equal(a, b) for a, b in zip(ref, flat(o3)))
print(f"dual chain, full-batch tiles: {t3:.3f} ms   outputs identical: {same}", flush=True)
