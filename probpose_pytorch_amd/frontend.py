"""Front end of the path: person boxes of an RGB frame -> network input crops on the GPU.

Counterpart of the reference's ``dataset.scale_box`` (dataset.py:71-90: ``image.crop(box)`` then
``.resize(image_size, PIL.Image.LANCZOS)`` and the keypoint rescale) followed by the
``v2.ToImage(); v2.ToDtype(torch.float32, scale=True)`` transform (dataset.py:107-112,
inference.py:74-82).  The pixel arithmetic is Pillow's; ``crop_resize`` reproduces it bit for bit
(csrc/pp_frontend.hip): the per-box coefficient tables are built on the host by the library with
Pillow's own double-precision expressions, the two resampling passes run in one HIP kernel.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib


def round_boxes(boxes_xywh) -> np.ndarray:
    """[x, y, w, h] boxes (dataset.py:74-80 passes (x, y, x+w, y+h) to Image.crop) -> the integer
    (x0, y0, x1, y1) Pillow crops: ``map(int, map(round, box))`` (Image.crop), i.e. Python's
    round-half-to-even on the float corner coordinates."""
    out = np.empty((len(boxes_xywh), 4), dtype=np.int32)
    for i, b in enumerate(boxes_xywh):
        x, y, w, h = (float(v) for v in b)
        out[i] = [int(round(x)), int(round(y)), int(round(x + w)), int(round(y + h))]
    return out


class FrontendPlan:
    """Host-built tables for one set of boxes and one output size (reusable across frames)."""

    def __init__(self, boxes_xyxy: np.ndarray, input_size: Sequence[int], device):
        L = _lib.lib()
        self.boxes = np.ascontiguousarray(boxes_xyxy, dtype=np.int32).reshape(-1, 4)
        self.n = int(self.boxes.shape[0])
        self.out_w, self.out_h = int(input_size[0]), int(input_size[1])
        bp = self.boxes.ctypes.data_as(C.c_void_p)
        nbytes = L.pp_frontend_plan_bytes(self.n, bp, self.out_w, self.out_h)
        if nbytes < 0:
            _lib.check(-1, "pp_frontend_plan_bytes")
        host = np.empty((max(int(nbytes), 4) // 4,), dtype=np.int32)
        nb, lds = C.c_int(0), C.c_longlong(0)
        _lib.check(L.pp_frontend_plan_build(self.n, bp, self.out_w, self.out_h, host.ctypes.data_as(C.c_void_p),
                                            C.byref(nb), C.byref(lds)), "pp_frontend_plan_build")
        self.host = host
        self.n_blocks, self.lds_bytes = int(nb.value), int(lds.value)
        self.dev = torch.from_numpy(host).to(device) if device is not None else None


def crop_resize(image: torch.Tensor, boxes_xywh, input_size: Sequence[int], plan: FrontendPlan = None) -> torch.Tensor:
    """image: (H, W, 3) uint8 RGB on the GPU; boxes [x, y, w, h] in pixels (floats allowed);
    input_size = [w, h].  Returns (n, 3, h, w) float32 in [0, 1] on the GPU: for every box exactly
    ``ToDtype(float32, scale=True)(ToImage()(image.crop(box).resize(input_size, LANCZOS)))``."""
    _lib.require_device(image)
    if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
        raise TypeError("image must be a (H, W, 3) uint8 tensor (RGB, channels last)")
    if image.stride(2) != 1 or image.stride(1) != 3:
        image = image.contiguous()
    if plan is None:
        plan = FrontendPlan(round_boxes(boxes_xywh), input_size, image.device)
    out = torch.empty((plan.n, 3, plan.out_h, plan.out_w), dtype=torch.float32, device=image.device)
    if plan.n == 0:
        return out
    with torch.cuda.device(image.device):
        rc = _lib.lib().pp_frontend_crop_resize(_lib.ptr(image), int(image.shape[1]), int(image.shape[0]),
                                                int(image.stride(0)), _lib.ptr(plan.dev), plan.n, plan.n_blocks,
                                                plan.lds_bytes, plan.out_w, plan.out_h, _lib.ptr(out),
                                                _lib.stream_ptr())
    _lib.check(rc, "pp_frontend_crop_resize")
    return out


def scale_box(image: torch.Tensor, bbox, image_size: Tuple[int, int], kps: np.ndarray):
    """Reference dataset.py:71-90 for one box: returns (crop (3, h, w) f32 in [0,1] on the GPU, kps)
    with the keypoints moved into the crop's pixel frame (kps is modified in place, like the
    reference; the keypoint rescale uses the UN-rounded box, dataset.py:87-89)."""
    crop = crop_resize(image, [bbox], image_size)[0]
    kps[:, 0] = (kps[:, 0] - bbox[0]) / bbox[2] * image_size[0]
    kps[:, 1] = (kps[:, 1] - bbox[1]) / bbox[3] * image_size[1]
    return crop, kps
