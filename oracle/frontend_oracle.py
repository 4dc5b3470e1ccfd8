"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU reference of the crop + LANCZOS-resize front end.

The reference's front end (dataset.py:71-90 ``scale_box``; inference.py:74-82) is three Pillow /
torchvision calls; the arithmetic lives in Pillow (third party, version 12.2.0 in this image; the
reference's requirements do not pin it).  Two oracles:

* ``scale_box_pil``: the reference's own call sequence on Pillow, i.e. the real thing
  (``image.crop(...)``, ``.resize(size, PIL.Image.LANCZOS)``), then ``v2.ToDtype(torch.float32, scale=True)``
  restated as ``float32(u8) * float32(1/255)`` (torchvision's to_dtype_image: ``image.to(dtype).mul_(1.0 /
  255)``; torchvision is not installed, so that last step is unpinned by a library run).  Parity of the
  pixel values (uint8) is pinned on Pillow itself.
* ``resize_lanczos_u8`` / ``crop_zero_pad`` / ``precompute_coeffs``: a numpy restatement of Pillow's
  ``ImagingResample`` 8-bit path (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
  ImagingResampleHorizontal_8bpc / Vertical_8bpc, PRECISION_BITS = 22) and of ``Image.crop``'s rounding and
  zero padding; checked against Pillow in tests/test_frontend.py and used there to validate the
  host-side plan builder of the library without a GPU.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def lanczos_filter(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size).
    Returns ksize, bounds [out,2] int, kk [out,ksize] int (22-bit fixed point)."""
    in0, in1 = 0.0, float(np.float32(in_size))
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [lanczos_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            kk[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def resize_lanczos_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """(h, w, 3) uint8 -> (out_h, out_w, 3) uint8, = PIL resize((out_w, out_h), LANCZOS)."""
    ih, iw, _ = img.shape
    _, bh, kh = precompute_coeffs(iw, out_w)
    _, bv, kv = precompute_coeffs(ih, out_h)
    cur = img.astype(np.int64)
    half = 1 << (PRECISION_BITS - 1)
    if out_w != iw:
        yf, yl = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
        tmp = np.zeros((yl - yf, out_w, 3), np.int64)
        for xx in range(out_w):
            xmin, xmax = (int(v) for v in bh[xx])
            ss = half + (cur[yf:yl, xmin:xmin + xmax] * kh[xx, :xmax][None, :, None]).sum(1)
            tmp[:, xx] = np.clip(ss >> PRECISION_BITS, 0, 255)
        bv = bv.copy()
        bv[:, 0] -= yf
        cur = tmp
    if out_h != ih:
        out = np.zeros((out_h, cur.shape[1], 3), np.int64)
        for yy in range(out_h):
            ymin, ymax = (int(v) for v in bv[yy])
            ss = half + (cur[ymin:ymin + ymax] * kv[yy, :ymax][:, None, None]).sum(0)
            out[yy] = np.clip(ss >> PRECISION_BITS, 0, 255)
        cur = out
    return cur.astype(np.uint8)


def round_box(bbox_xywh):
    """Image.crop: x0, y0, x1, y1 = map(int, map(round, box)) on (x, y, x+w, y+h)."""
    x, y, w, h = (float(v) for v in bbox_xywh)
    return int(round(x)), int(round(y)), int(round(x + w)), int(round(y + h))


def crop_zero_pad(img: np.ndarray, box_xyxy) -> np.ndarray:
    x0, y0, x1, y1 = box_xyxy
    out = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0), 3), np.uint8)
    H, W, _ = img.shape
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, W), min(y1, H)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = img[sy0:sy1, sx0:sx1]
    return out


def scale_box_numpy(img: np.ndarray, bbox_xywh, image_size) -> np.ndarray:
    """Restatement: (3, h, w) float32 in [0,1]."""
    crop = crop_zero_pad(img, round_box(bbox_xywh))
    res = resize_lanczos_u8(crop, int(image_size[0]), int(image_size[1]))
    return (res.astype(np.float32) * np.float32(1.0 / 255.0)).transpose(2, 0, 1).copy()


def scale_box_pil(img: np.ndarray, bbox_xywh, image_size) -> np.ndarray:
    """The reference's call sequence on Pillow itself (dataset.py:75-86, :107-112)."""
    import PIL.Image
    image = PIL.Image.fromarray(img, "RGB")
    cropped = image.crop((bbox_xywh[0], bbox_xywh[1], bbox_xywh[0] + bbox_xywh[2], bbox_xywh[1] + bbox_xywh[3]))
    scaled = cropped.resize(tuple(int(v) for v in image_size), resample=PIL.Image.LANCZOS)
    arr = np.asarray(scaled, dtype=np.uint8)
    return (arr.astype(np.float32) * np.float32(1.0 / 255.0)).transpose(2, 0, 1).copy()
