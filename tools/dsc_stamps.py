#!/usr/bin/env python3
"""Diagnostic: phase cycle shares of the screened decode kernel (-DPP_DSC_STAMPS build, lib/diag/libpp_dsc_stamps.so).
usage: dsc_stamps.py [B] [peaked|uniform]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import probpose_oracle as orc
from probpose_pytorch_amd.heatmap import oks_tap_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "probpose_pytorch_amd", "lib", "diag", "libpp_dsc_stamps.so"))
L.pp_decode_f32.restype = C.c_int
vp, i, d = C.c_void_p, C.c_int, C.c_double
L.pp_decode_f32.argtypes = [vp] * 5 + [i] * 4 + [vp, vp] + [d] * 4 + [vp] * 8 + [i, vp]
B, K, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 17, 64, 48
kind = sys.argv[2] if len(sys.argv) > 2 else "peaked"
taps, radius = oks_tap_table(K, H, W, orc.COCO17_SIGMAS)
taps, radius = torch.from_numpy(taps).cuda(), torch.from_numpy(radius).cuda()
hm = torch.from_numpy(orc.synthetic_heatmaps(B, K, H, W, 4321, kind)).cuda()
locs = torch.zeros((B * K * 2 + B * K * 16 + 64,), dtype=torch.float32, device="cuda")
kpts = torch.zeros((B, K, 2), dtype=torch.float64, device="cuda")
scores = torch.zeros((B, K), device="cuda")
for _ in range(3):
    rc = L.pp_decode_f32(hm.data_ptr(), None, None, None, None, B, K, H, W, taps.data_ptr(), radius.data_ptr(),
                         47.0, 63.0, 192.0, 256.0, kpts.data_ptr(), scores.data_ptr(), locs.data_ptr(), None, None,
                         None, None, None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
st = locs[2 * B * K: 2 * B * K + B * K * 16].view(torch.int64).cpu().numpy().reshape(B * K, 8)
names = ["load+A", "screen 1", "screen 2", "candidates", "argmax+nbrs", "total"]
for r in sorted(set(st[:, 6].tolist())):
    sel = st[st[:, 6] == r]
    print(f"radius {r}: {len(sel)} maps, candidates mean {sel[:, 7].mean():.1f} max {sel[:, 7].max()}  " +
          "  ".join(f"{n}: {sel[:, j].mean():7.0f}" for j, n in enumerate(names)))
