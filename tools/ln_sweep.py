#!/usr/bin/env python3
"""Time pp_layernorm standalone (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probpose_pytorch_amd import _lib, ops

if "--lib" in sys.argv:      # A/B an experimental build (tools/build_exp.sh <name> [flags]); one library per process
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "probpose_pytorch_amd",
                                 "lib", "exp", sys.argv[sys.argv.index("--lib") + 1])
    print("library:", _lib.LIB_PATH)

for rows, C in ((12288, 768), (49152, 1024), (12288, 384), (55296, 1280)):
    x = torch.randn((rows, C), device="cuda")
    g = torch.randn((C,), device="cuda")
    b = torch.randn((C,), device="cuda")
    out = torch.empty((rows, C), dtype=torch.bfloat16, device="cuda")
    for _ in range(5):
        ops.layernorm(x, g, b, 1e-6, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.layernorm(x, g, b, 1e-6, out)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / 50 * 1e3
    print(f"rows={rows} C={C}: {t:.1f} us  {rows * C * 6 / t / 1e3:.0f} GB/s")
