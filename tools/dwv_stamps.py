#!/usr/bin/env python3
"""Diagnostic: phase cycles of the wave-per-map decode kernel (-DPP_DWV_STAMPS build) on the bench model's heatmaps."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from probpose_pytorch_amd.heatmap import oks_tap_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "probpose_pytorch_amd", "lib", "diag", "libpp_dwv_stamps.so"))
L.pp_decode_f32.restype = C.c_int
vp, i, d = C.c_void_p, C.c_int, C.c_double
L.pp_decode_f32.argtypes = [vp] * 5 + [i] * 4 + [vp, vp] + [d] * 4 + [vp] * 8 + [i, vp]
cfg = bench.CONFIGS["vit_b"]
dev = torch.device("cuda", 0)
model, codec, sd = bench.build(cfg, torch.bfloat16, dev)
B = 64
x = torch.rand((B, 3, 256, 192), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
with torch.no_grad():
    hm = model(x)[0].float().contiguous()
_, K, H, W = hm.shape
taps, radius = oks_tap_table(K, H, W, bench.sigmas_for(K))
taps, radius = torch.from_numpy(taps).cuda(), torch.from_numpy(radius).cuda()
locs = torch.zeros((B * K * 2 + B * K * 16 + 64,), dtype=torch.float32, device="cuda")
kpts = torch.zeros((B, K, 2), dtype=torch.float64, device="cuda")
scores = torch.zeros((B, K), device="cuda")
L.pp_decode_workspace_bytes.restype = C.c_size_t
ws = torch.zeros((L.pp_decode_workspace_bytes(B, K, H, W),), dtype=torch.uint8, device="cuda")
for _ in range(3):
    rc = L.pp_decode_f32(hm.data_ptr(), None, None, None, None, B, K, H, W, taps.data_ptr(), radius.data_ptr(),
                         47.0, 63.0, 192.0, 256.0, kpts.data_ptr(), scores.data_ptr(), locs.data_ptr(), None, None,
                         None, None, ws.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
st = locs[2 * B * K: 2 * B * K + B * K * 16].view(torch.int64).cpu().numpy().reshape(B * K, 8)
names = ["load+stats", "passes", "scan", "exact", "neighbours+finalize", "total"]
for label, sel in (("<= 64 candidates", st[(st[:, 7] <= 64) & (st[:, 5] > 0)]), ("> 64 candidates", st[st[:, 7] > 64])):
    if len(sel):
        print(f"{label}: {len(sel)} maps, candidates mean {sel[:, 7].mean():.1f} max {sel[:, 7].max()}  " +
              "  ".join(f"{n}: {sel[:, j].mean():7.0f}" for j, n in enumerate(names)) + f"  (max total {sel[:, 5].max()})")
