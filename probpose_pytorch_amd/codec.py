"""Mirror of the hot-path part of the reference's ``probpose/codec.py``:
``ProbMap`` (codec.py:73-239, decode side) and ``Codec`` (codec.py:242-267).

``Codec.decode`` keeps the reference return structure but never copies the
heatmaps to the host: one fused HIP launch decodes the whole batch on the
device and only B*K*7 numbers cross PCIe.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .heatmap import decode_on_device
from .util import to_numpy


class ProbMap:
    """Reference codec.py:117-136 (constructor) and :214-239 (decode).

    ``input_size`` is [w, h], ``heatmap_size`` is [W, H]."""

    def __init__(self, input_size, heatmap_size, sigmas, sigma: float = 2.0,
                 radius_factor: float = 0.0546875, blur_kernel_size: int = 11,
                 increase_sigma_with_padding=False) -> None:
        self.input_size = input_size
        self.heatmap_size = heatmap_size
        self.radius_factor = radius_factor
        self.blur_kernel_size = blur_kernel_size
        self.scale_factor = ((np.array(input_size) - 1) / (np.array(heatmap_size) - 1)).astype(np.float32)
        self.increase_sigma_with_padding = increase_sigma_with_padding
        self.sigmas = sigmas
        self.sigma = sigma

    # -- device side ------------------------------------------------------
    def decode_device(self, heatmaps: torch.Tensor, aux=None) -> dict:
        """(B,K,H,W) device heatmaps -> dict of device tensors (kpts f64, scores f32, ...)."""
        W, H = self.heatmap_size
        return decode_on_device(heatmaps, self.sigmas, den=(W - 1, H - 1),
                                input_size=self.input_size, aux=aux)

    # -- reference surface ------------------------------------------------
    def decode(self, encoded):
        """(K,H,W) [or (B,K,H,W)] heatmaps -> keypoints (1,K,2) f64 [or (B,K,2)],
        scores (1,K) f32 [or (B,K)].  Reference codec.py:214-239; the input is
        never modified (the reference copies it, :228)."""
        _lib.require_device()
        t = torch.from_numpy(np.ascontiguousarray(encoded)) if isinstance(encoded, np.ndarray) else encoded
        if t.ndim == 3:
            t = t[None]
        if t.ndim != 4:
            raise AssertionError(f"Invalid shape {tuple(t.shape)}")
        if not t.is_cuda:
            t = t.cuda(non_blocking=True)
        out = self.decode_device(t)
        return out["kpts"].cpu().numpy(), out["scores"].cpu().numpy()

    def _two_s(self) -> np.ndarray:
        """2 * s_k in float64, s_k as reference codec.py:61-66: clip((2 sigma_k)^2 * sqrt(H/1.25 * W/1.25) * 2,
        0.55, 3.0), replaced by ``self.sigma`` when that is set and positive."""
        W, H = self.heatmap_size
        bbox_area = np.sqrt(H / 1.25 * W / 1.25)
        s = np.clip((np.asarray(self.sigmas, dtype=np.float64) * 2) ** 2 * bbox_area * 2, 0.55, 3.0)
        if self.sigma is not None and self.sigma > 0:
            s = np.full_like(s, float(self.sigma))
        return 2 * s

    def encode_device(self, keypoints, keypoints_visible=None):
        """Batched target generation on the GPU: keypoints (B, K, 2) in input-image pixels (numpy or tensor),
        visibility (B, K) or None -> (heatmaps (B, K, H, W) f32, keypoint_weights (B, K) f32) device tensors.
        Row b equals the reference ``encode(keypoints[b:b+1], ...)`` (codec.py:138-212, single instance)."""
        _lib.require_device()
        kp = np.asarray(keypoints.cpu() if isinstance(keypoints, torch.Tensor) else keypoints)
        B, K = kp.shape[:2]
        vis = (np.ones((B, K), dtype=np.float32) if keypoints_visible is None else
               np.asarray(keypoints_visible.cpu() if isinstance(keypoints_visible, torch.Tensor) else keypoints_visible))
        hm_kp = (kp[..., :2] / self.scale_factor).astype(np.float32)      # codec.py:178, float32 like the reference
        W, H = self.heatmap_size
        dev = torch.device("cuda", torch.cuda.current_device())
        d_kp = torch.from_numpy(np.ascontiguousarray(hm_kp)).to(dev)
        d_vis = torch.from_numpy(np.ascontiguousarray(vis, dtype=np.float32)).to(dev)
        d_s = torch.from_numpy(self._two_s()).to(dev)
        if d_s.numel() != K:
            raise ValueError(f"{K} keypoints but {d_s.numel()} sigmas")
        heat = torch.empty((B, K, H, W), dtype=torch.float32, device=dev)
        wts = torch.empty((B, K), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().pp_encode_probmaps(_lib.ptr(d_kp), _lib.ptr(d_vis), _lib.ptr(d_s), B, K, H, W,
                                               _lib.ptr(heat), _lib.ptr(wts), _lib.stream_ptr())
        _lib.check(rc, "pp_encode_probmaps")
        return heat, wts

    def encode(self, keypoints, keypoints_visible=None, id_similarity=0.0, keypoints_visibility=None) -> dict:
        """Reference codec.py:138-212: single-instance keypoints (1, K, D) in input-image pixels -> the target
        dict (heatmaps (K, H, W) f32, keypoint_weights (1, K), annotated, in_image, keypoints_scaled,
        heatmap_keypoints, identification_similarity); the maps are generated by the HIP kernel."""
        assert keypoints.shape[0] == 1, f"{self.__class__.__name__} only support single-instance keypoint encoding"
        if keypoints_visible is None:
            keypoints_visible = np.ones(keypoints.shape[:2], dtype=np.float32)
        heat, wts = self.encode_device(keypoints, keypoints_visible)
        weights = keypoints_visible.copy()
        weights[...] = wts.cpu().numpy().astype(weights.dtype)
        in_image = np.logical_and(keypoints[:, :, 0] >= 0, keypoints[:, :, 0] < self.input_size[0])
        in_image = np.logical_and(in_image, keypoints[:, :, 1] >= 0)
        in_image = np.logical_and(in_image, keypoints[:, :, 1] < self.input_size[1])
        return dict(heatmaps=heat[0].cpu().numpy(), keypoint_weights=weights, annotated=keypoints_visible > 0,
                    in_image=in_image, keypoints_scaled=keypoints, heatmap_keypoints=keypoints / self.scale_factor,
                    identification_similarity=id_similarity)


def gaussian_blur_taps(ksize: int) -> np.ndarray:
    """The float32 coefficients cv2.GaussianBlur(img_f32, (k, k), 0) filters with (reference codec.py:310):
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8, t_i = float32(exp(-0.5 x_i^2 / sigma^2)), normalised by the float64 sum
    of the float32 values (OpenCV getGaussianKernel, CV_32F).  Restated from OpenCV's published source; cv2 itself is
    not importable here (parity unpinned)."""
    if ksize % 2 != 1 or ksize < 1:
        raise AssertionError("kernel size must be odd")
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    t = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)
    return (t.astype(np.float64) * (1.0 / t.astype(np.float64).sum())).astype(np.float32)


class ArgMaxProbMap(ProbMap):
    """Reference codec.py:377-543: same targets as ProbMap (encode), decoding by raw arg-max + DARK-UDP refinement
    (codec.py:515-543) -- the decoder the reference's loss calls (train.py:47-48).  ``decode`` runs one fused HIP
    launch over the whole batch (csrc/pp_dark.hip)."""

    def __init__(self, input_size, heatmap_size, sigmas=None, sigma: float = -1, radius_factor: float = 0.0546875,
                 blur_kernel_size: int = 11, increase_sigma_with_padding=False) -> None:
        super().__init__(input_size, heatmap_size, sigmas, sigma=sigma, radius_factor=radius_factor,
                         blur_kernel_size=blur_kernel_size, increase_sigma_with_padding=increase_sigma_with_padding)

    def decode_device(self, heatmaps: torch.Tensor, aux=None) -> dict:
        """(B,K,H,W) f32 device heatmaps -> dict(kpts (B,K,2) f64, scores (B,K) f32, locs (B,K,2) f32) on the device."""
        _lib.require_device(heatmaps)
        hm = heatmaps.detach()
        if hm.dtype != torch.float32 or not hm.is_contiguous():
            hm = hm.float().contiguous()
        B, K, H, W = hm.shape
        taps = np.ascontiguousarray(gaussian_blur_taps(int(self.blur_kernel_size)))
        kpts = torch.empty((B, K, 2), dtype=torch.float64, device=hm.device)
        scores = torch.empty((B, K), dtype=torch.float32, device=hm.device)
        locs = torch.empty((B, K, 2), dtype=torch.float32, device=hm.device)
        with torch.cuda.device(hm.device):
            rc = _lib.lib().pp_dark_decode_f32(_lib.ptr(hm), B, K, H, W, taps.ctypes.data, taps.shape[0],
                                               float(self.input_size[0]), float(self.input_size[1]), _lib.ptr(kpts),
                                               _lib.ptr(scores), _lib.ptr(locs), _lib.stream_ptr())
        _lib.check(rc, "pp_dark_decode_f32")
        out = dict(kpts=kpts, scores=scores, locs=locs)
        if aux is not None:       # Codec.decode's pass-through of the four scalar heads (codec.py:254-261)
            a = [t.reshape(B, K).float() for t in aux]
            out["aux"] = torch.stack(a[:3])
            out["err"] = a[3].double() / float(np.sqrt(H * H + W * W))
        return out

    def decode(self, encoded):
        """(K,H,W) [or (B,K,H,W)] heatmaps -> keypoints (1,K,2) f64 [or (B,K,2)], scores (1,K) f32 [or (B,K)]."""
        _lib.require_device()
        t = torch.from_numpy(np.ascontiguousarray(encoded)) if isinstance(encoded, np.ndarray) else encoded
        if t.ndim == 3:
            t = t[None]
        if t.ndim != 4:
            raise AssertionError(f"Invalid shape {tuple(t.shape)}")
        if not t.is_cuda:
            t = t.cuda(non_blocking=True)
        out = self.decode_device(t)
        return out["kpts"].cpu().numpy(), out["scores"].cpu().numpy()


class Codec:
    """Reference codec.py:242-267."""

    def __init__(self, probmap):
        self.probmap = probmap

    def decode_device(self, pred) -> dict:
        """Fused decode that leaves everything on the device (used by the
        data-parallel all-gather): kpts (B,K,2) f64, scores (B,K) f32,
        aux (3,B,K) f32 = prob/vis/oks, err (B,K) f64 (already / diagonal)."""
        heatmaps, probabilities, visibilities, oks, errors = pred
        _lib.require_device(heatmaps)
        return self.probmap.decode_device(heatmaps, aux=(probabilities, visibilities, oks, errors))

    def decode(self, pred):
        """5-tuple (heatmaps, prob, vis, oks, err) -> ((kpts (B,K,2) f64, scores (B,K) f32),
        prob (B,1,K) f32, vis, oks, err (B,1,K) f64).  Reference codec.py:249-263
        (which only works for B == 1; B > 1 is the stack of per-crop results)."""
        heatmaps = pred[0]
        B, K = heatmaps.shape[0], heatmaps.shape[1]
        pred = tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() if isinstance(p, np.ndarray)
                     else p for p in pred)
        out = self.decode_device(pred)
        # one packed D2H instead of the reference's full-heatmap to_numpy (util.py:6-12)
        kpts = out["kpts"].cpu().numpy()
        scores = out["scores"].cpu().numpy()
        aux = out["aux"].cpu().numpy()
        err = out["err"].cpu().numpy()
        return ((kpts, scores), aux[0].reshape(B, 1, K), aux[1].reshape(B, 1, K),
                aux[2].reshape(B, 1, K), err.reshape(B, 1, K))

    def decode_heatmap(self, heatmaps):
        """Reference codec.py:265-267."""
        return self.probmap.decode(heatmaps if isinstance(heatmaps, torch.Tensor) else to_numpy(heatmaps))

    def encode(self, *args, **kwargs):
        return self.probmap.encode(*args, **kwargs)
