#!/usr/bin/env python3
"""Diagnostic: phase cycle shares of the fused decode kernel (-DPP_DEC_STAMPS build)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from probpose_pytorch_amd.heatmap import oks_tap_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "probpose_pytorch_amd", "lib", "diag", "libpp_dec_stamps.so"))
L.pp_decode_f32.restype = C.c_int
vp, i, d = C.c_void_p, C.c_int, C.c_double
L.pp_decode_f32.argtypes = [vp] * 5 + [i] * 4 + [vp, vp] + [d] * 4 + [vp] * 8 + [i, vp]
B, K, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 17, 64, 48
sig = np.array([.026, .025, .025, .035, .035, .079, .079, .072, .072, .062, .062, .107, .107, .087, .087, .089, .089])
taps, radius = oks_tap_table(K, H, W, sig)
taps, radius = torch.from_numpy(taps).cuda(), torch.from_numpy(radius).cuda()
hm = torch.rand((B, K, H, W), device="cuda")
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
names = ["load + barrier", "row+col passes", "argmax reduce", "finalize", "total"]
for r in sorted(set(st[:, 6].tolist())):
    sel = st[st[:, 6] == r]
    row = (sel[:, 7] >> 32).mean()
    bar = (sel[:, 7] & 0xffffffff).mean()
    print(f"radius {r}: {len(sel)} maps  " + "  ".join(f"{n}: {sel[:, j].mean():7.0f}" for j, n in enumerate(names)) +
          f"  [row pass {row:.0f}, barrier {bar:.0f}, col pass {sel[:, 1].mean() - row - bar:.0f}]")
print("span cycles", st[:, 5].max() + st[st[:, 5].argmax(), 4] - st[:, 5].min())
