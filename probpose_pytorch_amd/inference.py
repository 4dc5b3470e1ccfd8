"""Counterpart of the reference's ``probpose/inference.py`` (a ``__main__``-only CLI there).

The call sequence that defines the API contract (reference inference.py:61-112) is kept as a library
function, ``run_inference``: ``Codec(ProbMap(input_size, heatmap_size, sigmas))``, ``model(x)``,
``codec.decode(out)``; the CLI around it keeps the reference's flags.  Differences, all forced:

* weights are loaded as a ``state_dict`` with ``torch.load(..., weights_only=True)``; the reference's
  whole-module pickles (``weights_only=False``) execute code from the file and are refused unless
  ``--trust-pickle`` is given;
* ``--backbone`` defaults to the in-tree ViT: ``RadioBackbone`` needs a ``torch.hub`` download;
* the model runs in ``.eval()`` mode (the reference never calls it, i.e. runs train-mode BatchNorm);
* image I/O needs PIL; without ``--image`` a seeded synthetic crop is used.
"""
from __future__ import annotations

import argparse
from pathlib import Path

import numpy as np
import torch

from .backbone import ScratchViTBackbone
from .codec import Codec, ProbMap
from .head import ProbMapHead
from .model import ProbPoseModel

VARIANTS = {"vit_s": (384, 12, 12), "vit_b": (768, 12, 12), "vit_l": (1024, 24, 16), "vit_h": (1280, 32, 16)}


def default_pools(grid):
    """alt_head_kernel_sizes that reduce a (gh, gw) grid to 1x1 in three poolings (train.py:44 uses
    [(4,4),(2,2),(2,2)] for 24x24 -> here the same recipe for any grid divisible like 16x12 or 24x18)."""
    gh, gw = grid
    pools = []
    for _ in range(2):
        kh = 4 if gh % 4 == 0 and gh > 4 else (2 if gh % 2 == 0 and gh > 1 else (3 if gh % 3 == 0 and gh > 1 else 1))
        kw = 4 if gw % 4 == 0 and gw > 4 else (3 if gw % 3 == 0 and gw > 3 else (2 if gw % 2 == 0 and gw > 1 else 1))
        pools.append((kh, kw))
        gh, gw = gh // kh, gw // kw
    pools.append((gh, gw))
    return pools


def build_model(input_size, num_keypoints: int, variant: str = "vit_s", patch: int = 16):
    """input_size is [w, h] as on the reference command line; returns (model, heatmap_size [W, H])."""
    w, h = int(input_size[0]), int(input_size[1])
    C, depth, heads = VARIANTS[variant]
    grid = (h // patch, w // patch)
    backbone = ScratchViTBackbone((h, w), patch, embed_dim=C, depth=depth, num_heads=heads)
    head = ProbMapHead(C, num_keypoints, default_pools(grid), (256, 256), (4, 4), final_layer_kernel_size=1)
    return ProbPoseModel(backbone, head), (grid[1] * 4, grid[0] * 4)


def run_inference(model: ProbPoseModel, codec: Codec, image_tensor: torch.Tensor):
    """image_tensor (B,3,H,W) float32 in [0,1] on the GPU -> (raw 5-tuple, decoded predictions).
    Reference inference.py:82-107."""
    with torch.no_grad():
        output = model(image_tensor)
    return output, codec.decode(output)


def run_inference_on_boxes(model: ProbPoseModel, codec: Codec, frame: torch.Tensor, boxes_xywh):
    """Whole per-frame path on the GPU: ``frame`` (H, W, 3) uint8 RGB on the device and person boxes
    [x, y, w, h] -> crops (dataset.py:71-90 semantics, frontend.crop_resize) -> forward -> decode.
    Returns (raw 5-tuple, decoded predictions, keypoints in FRAME pixels (n, K, 2) float64): the inverse of
    the keypoint rescale of dataset.py:87-89, ``kpt / input_size * box_wh + box_xy``."""
    from . import frontend
    input_size = codec.probmap.input_size
    crops = frontend.crop_resize(frame, boxes_xywh, input_size)
    output, preds = run_inference(model, codec, crops)
    b = np.asarray(boxes_xywh, dtype=np.float64).reshape(-1, 4)
    kpts = np.asarray(preds[0][0], dtype=np.float64)
    in_wh = np.asarray(input_size, dtype=np.float64)
    frame_kpts = kpts / in_wh * b[:, None, 2:4] + b[:, None, 0:2]
    return output, preds, frame_kpts


def load_image(path: Path, input_size) -> torch.Tensor:
    """Reference inference.py:74-82: RGB, LANCZOS resize to input_size [w,h], scale to [0,1]."""
    import PIL.Image
    image = PIL.Image.open(path).convert("RGB").resize(tuple(input_size), PIL.Image.LANCZOS)
    arr = np.asarray(image, dtype=np.float32) * np.float32(1.0 / 255.0)   # v2.ToDtype(float32, scale=True)
    return torch.from_numpy(arr).permute(2, 0, 1).unsqueeze(0).contiguous()


def load_weights(model: ProbPoseModel, path: Path, model_type: str, trust_pickle: bool = False):
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:  # a whole-module pickle (reference train.py:171-180)
        if not trust_pickle:
            raise RuntimeError(f"{path} is not a plain state_dict (safe loader said: {e}); whole-module pickles run "
                               "code on load -- re-save as a state_dict or pass --trust-pickle") from e
        obj = torch.load(path, map_location="cpu", weights_only=False)
    sd = obj.state_dict() if hasattr(obj, "state_dict") else obj
    target = model.head if model_type == "head" else model
    return target.load_state_dict(sd)


def main(argv=None):
    p = argparse.ArgumentParser(description="Inference script for ProbPose (MI355X-native path)")
    p.add_argument("--model", type=Path, default=None, help="state_dict checkpoint (omit: seeded synthetic weights)")
    p.add_argument("--model_type", type=str, default="full", choices=["head", "full"])
    p.add_argument("--image", type=Path, default=None, help="input image (omit: seeded synthetic crop)")
    p.add_argument("--output", type=Path, default=None, help="folder for heatmap .npy dumps")
    p.add_argument("--backbone", type=str, default="vit_s", choices=sorted(VARIANTS))
    p.add_argument("--input_size", type=str, default="192,256", help="w,h")
    p.add_argument("--num_keypoints", type=int, default=17)
    p.add_argument("--sigma", type=float, default=0.05, help="per-keypoint OKS sigma (inference.py:72 uses one value)")
    p.add_argument("--bf16", action="store_true", help="bf16 MFMA instead of exact-fp32 MFMA")
    p.add_argument("--normalize", action="store_true", help="divide each dumped heatmap by its maximum")
    p.add_argument("--trust-pickle", action="store_true")
    args = p.parse_args(argv)
    input_size = tuple(map(int, args.input_size.split(",")))
    model, heatmap_size = build_model(input_size, args.num_keypoints, args.backbone)
    if args.model is not None:
        print("load:", load_weights(model, args.model, args.model_type, args.trust_pickle))
    else:
        from .synthetic import synthetic_model_state
        C, depth, _ = VARIANTS[args.backbone]
        model.load_state_dict(synthetic_model_state((input_size[1], input_size[0]), 16, C, depth, args.num_keypoints,
                                                    3, (256, 256), seed=0))
    model = model.to("cuda").eval()
    if args.bf16:
        model.set_compute_dtype(torch.bfloat16)
    codec = Codec(ProbMap(input_size, heatmap_size, np.array([args.sigma] * args.num_keypoints)))
    if args.image is not None:
        x = load_image(args.image, input_size)
    else:
        from .synthetic import synthetic_crops
        x = synthetic_crops(1, input_size[1], input_size[0], seed=1234)
    print("Input image shape:", tuple(x.shape))
    output, preds = run_inference(model, codec, x.to("cuda"))
    heatmaps = output[0][0].cpu().numpy()
    print("Output heatmap shape:", heatmaps.shape)
    if args.output is not None:
        args.output.mkdir(parents=True, exist_ok=True)
        for i, hm in enumerate(heatmaps):
            np.save(args.output / f"heatmap_{i}.npy", hm / hm.max() if args.normalize and hm.max() > 0 else hm)
    print("Predictions:", preds[0])
    print("Probabilities:", preds[1])
    print("Visibilities:", preds[2])
    print("OKS:", preds[3])
    print("Errors:", preds[4])
    return preds


if __name__ == "__main__":
    main()
