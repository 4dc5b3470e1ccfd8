"""Front end (crop + LANCZOS resize + [0,1] scaling; reference dataset.py:71-90, inference.py:74-82).

CPU tests: the numpy restatement == Pillow itself; the library's HOST plan builder (bounds + 22-bit
coefficient tables, no GPU needed) == the restatement's tables, and applying the plan in numpy == Pillow.
GPU tests (marked): the HIP kernel == Pillow, bit for bit, on seeded frames and boxes incl. boxes that
leave the frame, up- and down-scaling, identity sizes, 1-pixel boxes and .5 corners (round-half-even)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import frontend_oracle as fo

SIZE = (192, 256)   # [w, h]


def _frame(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = ((np.sin(xx / 17.0) + np.cos(yy / 23.0)) * 60 + 128).clip(0, 255).astype(np.uint8)
    base[: h // 2] = (base[: h // 2] // 4 + smooth[: h // 2, :, None] * 3 // 4).astype(np.uint8)
    return base


BOXES = [
    (10.0, 20.0, 100.0, 200.0),       # down-scale x, y
    (50.2, 60.7, 300.4, 420.9),       # fractional corners
    (0.5, 1.5, 192.0, 256.0),         # .5 corners: round-half-to-even
    (33.0, 44.0, 192.0, 256.0),       # identity size: copy
    (5.0, 5.0, 192.0, 100.0),         # vertical only
    (5.0, 5.0, 80.0, 256.0),          # horizontal only
    (-30.0, -40.0, 150.0, 300.0),     # leaves the frame top-left (zero padded)
    (500.0, 380.0, 200.0, 200.0),     # leaves the frame bottom-right
    (100.0, 100.0, 1.0, 1.0),         # one pixel, up-scale
    (0.0, 0.0, 640.0, 480.0),         # the whole frame
    (200.0, 100.0, 37.0, 51.0),       # up-scale
]


def test_restatement_equals_pillow():
    img = _frame(480, 640, 1)
    for b in BOXES:
        assert np.array_equal(fo.scale_box_numpy(img, b, SIZE), fo.scale_box_pil(img, b, SIZE)), b


def _plan(built_lib, boxes_xyxy, size):
    boxes = np.ascontiguousarray(boxes_xyxy, dtype=np.int32)
    bp = boxes.ctypes.data_as(C.c_void_p)
    n = boxes.shape[0]
    nbytes = built_lib.pp_frontend_plan_bytes(n, bp, size[0], size[1])
    assert nbytes > 0
    host = np.zeros((nbytes // 4,), dtype=np.int32)
    nb, lds = C.c_int(0), C.c_longlong(0)
    rc = built_lib.pp_frontend_plan_build(n, bp, size[0], size[1], host.ctypes.data_as(C.c_void_p), C.byref(nb),
                                          C.byref(lds))
    assert rc == 0, built_lib.pp_last_error()
    return host, nb.value, lds.value


def test_host_plan_tables_match_pillow_restatement(built_lib):
    """pp_frontend_plan_build runs on the host: its tables must be Pillow's, value for value."""
    boxes = np.array([fo.round_box(b) for b in BOXES], dtype=np.int32)
    host, n_blocks, lds = _plan(built_lib, boxes, SIZE)
    assert n_blocks >= len(BOXES) and 0 < lds <= 152 * 1024
    for c, box in enumerate(boxes):
        h = host[c * 16:(c + 1) * 16]
        cw, ch = box[2] - box[0], box[3] - box[1]
        assert (h[0], h[1], h[2], h[3]) == (box[0], box[1], cw, ch)
        for in_size, out_size, ks_slot, off_b, off_k in ((cw, SIZE[0], 4, 8, 9), (ch, SIZE[1], 5, 10, 11)):
            ksize, bounds, kk = fo.precompute_coeffs(int(in_size), out_size)
            assert h[ks_slot] == ksize
            assert np.array_equal(host[h[off_b]:h[off_b] + 2 * out_size].reshape(out_size, 2), bounds)
            assert np.array_equal(host[h[off_k]:h[off_k] + ksize * out_size].reshape(out_size, ksize), kk)
        assert h[12] == int(cw != SIZE[0]) and h[13] == int(ch != SIZE[1])
    # block table: every (box, row block) exactly once, in order
    bt = host[len(boxes) * 16: len(boxes) * 16 + 2 * n_blocks].reshape(n_blocks, 2)
    for c in range(len(boxes)):
        rb = host[c * 16 + 14]
        assert list(bt[bt[:, 0] == c][:, 1]) == list(range(0, SIZE[1], rb))


def test_plan_rejects_empty_and_oversized_boxes(built_lib):
    bad = np.array([[10, 10, 10, 50]], dtype=np.int32)
    assert built_lib.pp_frontend_plan_bytes(1, bad.ctypes.data_as(C.c_void_p), 192, 256) < 0
    assert b"empty box" in built_lib.pp_last_error()
    huge = np.array([[0, 0, 192, 256 * 400]], dtype=np.int32)   # 400x vertical down-scale: > 152 KB of LDS per row
    assert built_lib.pp_frontend_plan_bytes(1, huge.ctypes.data_as(C.c_void_p), 192, 256) < 0
    assert b"LDS" in built_lib.pp_last_error()


def test_round_boxes_is_pillows_rounding():
    from probpose_pytorch_amd.frontend import round_boxes
    got = round_boxes(BOXES)
    assert [tuple(r) for r in got] == [fo.round_box(b) for b in BOXES]
    assert tuple(round_boxes([(0.5, 1.5, 2.0, 1.0)])[0]) == (0, 2, 2, 2)   # half to even: 0.5 -> 0, 1.5 -> 2, 2.5 -> 2


@pytest.mark.gpu
def test_crop_resize_equals_pillow_bit_exact(built_lib):
    from probpose_pytorch_amd import frontend
    img = _frame(480, 640, 2)
    dev = torch.from_numpy(img).cuda()
    out = frontend.crop_resize(dev, BOXES, SIZE).cpu().numpy()
    assert out.shape == (len(BOXES), 3, SIZE[1], SIZE[0]) and out.dtype == np.float32
    for i, b in enumerate(BOXES):
        ref = fo.scale_box_pil(img, b, SIZE)
        assert np.array_equal(out[i], ref), (b, np.abs(out[i] - ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(192, 256), (288, 384), (96, 96)])
def test_crop_resize_random_boxes(built_lib, size):
    from probpose_pytorch_amd import frontend
    rng = np.random.default_rng(7)
    img = _frame(720, 1280, 3)
    boxes = [(float(rng.uniform(-50, 1200)), float(rng.uniform(-50, 650)), float(rng.uniform(8, 700)),
              float(rng.uniform(8, 900))) for _ in range(24)]
    out = frontend.crop_resize(torch.from_numpy(img).cuda(), boxes, size).cpu().numpy()
    for i, b in enumerate(boxes):
        assert np.array_equal(out[i], fo.scale_box_pil(img, b, size)), b


@pytest.mark.gpu
def test_crop_resize_strided_frame_empty_batch_and_scale_box(built_lib):
    from probpose_pytorch_amd import frontend
    img = _frame(300, 400, 4)
    wide = torch.zeros((300, 512, 3), dtype=torch.uint8, device="cuda")
    wide[:, :400] = torch.from_numpy(img).cuda()
    view = wide[:, :400]                                   # row stride 1536 B, not 1200
    b = (20.0, 30.0, 150.0, 220.0)
    assert np.array_equal(frontend.crop_resize(view, [b], SIZE)[0].cpu().numpy(), fo.scale_box_pil(img, b, SIZE))
    assert frontend.crop_resize(view, [], SIZE).shape == (0, 3, 256, 192)
    kps = np.array([[50.0, 80.0], [170.0, 250.0]], dtype=np.float32)
    crop, k2 = frontend.scale_box(view, list(b), SIZE, kps.copy())
    assert crop.shape == (3, 256, 192)
    np.testing.assert_allclose(k2, [[(50 - 20) / 150 * 192, (80 - 30) / 220 * 256],
                                    [(170 - 20) / 150 * 192, (250 - 30) / 220 * 256]], rtol=1e-6)
    with pytest.raises(TypeError):
        frontend.crop_resize(view.float(), [b], SIZE)
