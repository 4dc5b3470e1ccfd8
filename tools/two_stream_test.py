#!/usr/bin/env python3
"""Experiment: one 64-crop forward vs two 32-crop forwards on two streams (independent plans)."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from probpose_pytorch_amd import ops
from probpose_pytorch_amd.synthetic import synthetic_crops

ops.AUTOTUNE = True
cfg = dict(bench.CONFIGS["vit_b"])
dev = torch.device("cuda", 0)
model, codec, _ = bench.build(cfg, torch.bfloat16, dev)
model_b = copy.deepcopy(model)
x = synthetic_crops(64, 256, 192).to(dev)
xa, xb = x[:32].contiguous(), x[32:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    return codec.decode_device(model(x))


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = codec.decode_device(model(xa))
    with torch.cuda.stream(s2):
        b = codec.decode_device(model_b(xb))
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    return a, b


def timeit(fn, iters=20):
    with torch.no_grad():
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters


print("one 64-crop forward  :", round(timeit(one), 3), "ms")
print("two 32-crop forwards :", round(timeit(two), 3), "ms")
