#!/usr/bin/env python3
"""A/B the decode forms (pp_decode_f32 flags: default wave-per-map / workgroup-per-map screened / all-pixel float64) in ONE process:
HIP-event time of the bare C call on resident buffers, interleaved rounds, median."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import probpose_oracle as orc
from probpose_pytorch_amd import _lib
from probpose_pytorch_amd.heatmap import oks_tap_table

L = _lib.lib()
for (B, K, H, W, sig) in ((64, 17, 64, 48, orc.COCO17_SIGMAS), (256, 17, 64, 48, orc.COCO17_SIGMAS),
                          (1024, 17, 64, 48, orc.COCO17_SIGMAS),
                          (128, 133, 96, 72, np.random.default_rng(133).uniform(0.02, 0.11, 133))):
    taps, radius = oks_tap_table(K, H, W, sig)
    taps, radius = torch.from_numpy(taps).cuda(), torch.from_numpy(radius).cuda()
    hm = torch.from_numpy(orc.synthetic_heatmaps(min(B, 64), K, H, W, 4321, "peaked")).cuda()
    hm = hm.repeat((B + hm.shape[0] - 1) // hm.shape[0], 1, 1, 1)[:B].contiguous()
    kpts = torch.zeros((B, K, 2), dtype=torch.float64, device="cuda")
    scores = torch.zeros((B, K), device="cuda")
    locs = torch.zeros((B, K, 2), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.pp_decode_workspace_bytes.restype = C.c_size_t
    nws = L.pp_decode_workspace_bytes(B, K, H, W)
    ws = torch.zeros((max(nws, 16),), dtype=torch.uint8, device="cuda")
    MODES = {"default": 0, "wave-per-map": _lib.DECODE_WAVE, "wg screened": _lib.DECODE_NO_WAVE | _lib.DECODE_SCREEN,
             "all-pixel f64": _lib.DECODE_ALL_PIXEL}

    def call(flags=0):
        rc = L.pp_decode_f32(hm.data_ptr(), None, None, None, None, B, K, H, W, taps.data_ptr(), radius.data_ptr(),
                             float(W - 1), float(H - 1), float(4 * W), float(4 * H), kpts.data_ptr(), scores.data_ptr(),
                             locs.data_ptr(), None, None, None, None, ws.data_ptr() if nws else None, flags, st)
        assert rc == 0, L.pp_last_error()

    res = {m: [] for m in MODES}
    for rnd in range(5):
        for mode, fl in MODES.items():
            call_ = call
            call = lambda fl=fl: call_(fl)
            for _ in range(3):
                call()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                call()
            e.record()
            e.synchronize()
            res[mode].append(s.elapsed_time(e) / 20 * 1e3)
            call = call_
    byts = B * K * H * W * 4
    for mode, t in res.items():
        m = float(np.median(t))
        print(f"B={B:5d} K={K:3d} {H}x{W}  {mode:14s} {m:8.1f} us  {byts / m / 1e3:8.1f} GB/s  ({byts / m / 1e3 / 8000:.3f} of 8 TB/s)")
