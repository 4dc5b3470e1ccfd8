"""CPU oracle for the ProbPose forward + decode path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (numpy / scipy /
plain torch fp32) of what the reference computes on its hot path.  It is the
*checker* used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``probpose_pytorch_amd/``
imports it; the product path is HIP-only and fails loudly without the
extension.

Parity pinning (see DESIGN.md "Oracle"):
  * decode  (rows C2-C4, D1-D4): pinned against the imported reference
    (``probpose.codec.Codec.decode`` / ``get_heatmap_expected_value``) through
    ``tests/golden/decode_*.npz`` minted by ``tests/golden/make_goldens.py``.
  * head    (rows H1-H6): pinned against the imported reference
    ``probpose.head.ProbMapHead`` through ``tests/golden/head_*.npz``.
  * Sparsemax (row H4, normalize != None): sparsemax==0.1.9 is not installed and
    not vendored -> PARITY UNPINNED; ``sparsemax_lastdim`` restates the published
    algorithm and is property-checked (simplex, support rule, KKT) in the tests.
  * backbone (rows B1-B6): the arithmetic lives in timm==1.0.15, which is not
    installed and not vendored -> PARITY UNPINNED by the reference; the
    restatement follows timm's published VisionTransformer semantics and is
    self-checked (explicit softmax vs SDPA, fp64 vs fp32).

Every function cites the reference file:line it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

COCO17_SIGMAS = np.array(
    [.026, .025, .025, .035, .035, .079, .079, .072, .072, .062, .062,
     .107, .107, .087, .087, .089, .089])


# --------------------------------------------------------------------------
# D2: OKS kernels                                   probpose/heatmap.py:170-194
# --------------------------------------------------------------------------
def oks_variance_and_radius(H: int, W: int, sigma_k) -> Tuple[float, int]:
    """``s`` (the Gaussian *variance*) and the window radius of keypoint k.

    heatmap.py:171 (bbox_area), :176-179 (vars, s, clip, radius).  ``sigma_k``
    is kept as the numpy scalar taken from the caller's array so a float32
    ``sigmas`` array rounds ``(2*sigma)**2`` in float32 exactly like the
    reference does before the promotion to float64 by ``bbox_area``.
    """
    bbox_area = np.sqrt(H / 1.25 * W / 1.25)
    var = (sigma_k * 2) ** 2
    s = np.clip(var * bbox_area * 2, 0.55, 3.0)
    radius = int(np.ceil(s * 3))
    return s, radius


def oks_kernels(K: int, H: int, W: int, sigmas: np.ndarray):
    """List of K float64 (1, d, d) normalised kernels.  heatmap.py:170-194."""
    out = []
    for k in range(K):
        s, r = oks_variance_and_radius(H, W, sigmas[k])
        ax = np.arange(2 * r + 1) - r
        gx, gy = np.meshgrid(ax, ax)
        dist = np.sqrt(gx ** 2 + gy ** 2)          # :187 (sqrt, squared again below)
        kern = np.exp(-(dist ** 2) / (2 * s))      # :188
        kern = kern / kern.sum()                   # :189
        out.append(kern.reshape(1, 2 * r + 1, 2 * r + 1))
    return out


# --------------------------------------------------------------------------
# D2c / D2t: the two convolution back-ends       heatmap.py:338-364, :196-288
# --------------------------------------------------------------------------
def convolve_reflect_scipy(maps: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """scipy back-end.  maps (B,H,W) f32, kern (1,d,d) f64.  heatmap.py:361-362."""
    from scipy.ndimage import convolve
    return convolve(maps, kern, mode="reflect")


def convolve_reflect_torch64(maps: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """float64 torch back-end: half-sample-symmetric pad (edge pixel repeated:
    ``d c b a | a b c d | d c b a``) + true convolution.  heatmap.py:196-288,
    call site :340-359.  Returns float64 (the caller stores into a float32
    array, :364)."""
    x = torch.from_numpy(np.ascontiguousarray(maps)).to(torch.float64)[:, None]
    d = kern.shape[-1]
    p = d // 2
    if p > 0:
        x = torch.cat([x[..., :p, :].flip(-2), x, x[..., x.shape[-2] - p:, :].flip(-2)], dim=-2)
        x = torch.cat([x[..., :, :p].flip(-1), x, x[..., :, x.shape[-1] - p:].flip(-1)], dim=-1)
    # the reference flips the kernel twice (call site + inside the helper) and
    # conv2d is a correlation, so the net operation is a correlation with the
    # (symmetric) kernel -- identical to a true convolution for these kernels.
    w = torch.from_numpy(np.ascontiguousarray(kern)).to(torch.float64).reshape(1, 1, d, d)
    return F.conv2d(x, w)[:, 0].numpy()


# --------------------------------------------------------------------------
# D3: sub-pixel refinement                          heatmap.py:114-167
# --------------------------------------------------------------------------
def subpixel_refine(conv: np.ndarray, locs: np.ndarray) -> np.ndarray:
    """conv (M,H,W) f32 convolved maps, locs (M,2) f32 integer (x,y) peaks."""
    M, H, W = conv.shape
    xi = locs[:, 0].astype(np.int32)
    yi = locs[:, 1].astype(np.int32)
    inner = (xi > 0) & (xi < W - 1) & (yi > 0) & (yi < H - 1)     # :120-125
    out = locs.copy()
    idx = np.nonzero(inner)[0]
    if idx.size:
        x, y = xi[idx], yi[idx]
        c = conv[idx, y, x]
        xp, xm = conv[idx, y, x + 1], conv[idx, y, x - 1]
        yp, ym = conv[idx, y + 1, x], conv[idx, y - 1, x]
        dx = (xp - xm) / 2.0                                     # :136-139
        dy = (yp - ym) / 2.0                                     # :140-143
        dxx = xp + xm - 2 * c                                    # :144-148
        dyy = yp + ym - 2 * c                                    # :149-153
        dxx = np.where(dxx != 0, dxx, 1e-6)                      # :156
        dyy = np.where(dyy != 0, dyy, 1e-6)                      # :157
        out[idx, 0] += -dx / dxx                                 # :160,164
        out[idx, 1] += -dy / dyy                                 # :161,165
    return out


# --------------------------------------------------------------------------
# D1 + D4: one crop                                 heatmap.py:291-395
# --------------------------------------------------------------------------
def heatmap_expected_value(heatmaps: np.ndarray, sigmas: np.ndarray,
                           backend: str = "scipy", return_heatmap: bool = False):
    """Single crop ``(K,H,W)`` float32 -> locs (K,2) f32, vals (K,) f32.

    Follows heatmap.py:323-395 for the ``ndim == 3`` case (B = 1); the
    reference raises for B > 1 (:362-364), see ``codec_decode`` below.
    """
    assert isinstance(heatmaps, np.ndarray) and heatmaps.ndim == 3
    K, H, W = heatmaps.shape
    kernels = oks_kernels(K, H, W, sigmas)
    conv = np.zeros_like(heatmaps)                                # :335
    for k in range(K):                                            # :338
        m = heatmaps[k][None]
        if backend == "torch":
            conv[k] = convolve_reflect_torch64(m, kernels[k])[0]
        else:
            conv[k] = convolve_reflect_scipy(m, kernels[k])[0]
    flat = np.argmax(conv.reshape(K, H * W), axis=1)              # :366-369
    yi, xi = np.unravel_index(flat, (H, W))
    locs = np.stack((xi, yi), axis=-1).astype(np.float32)         # :370
    locs = subpixel_refine(conv, locs)                            # :373
    vals = heatmaps[np.arange(K), yi, xi]                         # :375-379 (raw map)
    if return_heatmap:
        return locs, vals, conv
    return locs, vals


# --------------------------------------------------------------------------
# C4: ProbMap.decode                                 codec.py:214-239
# --------------------------------------------------------------------------
def probmap_decode(heatmaps: np.ndarray, input_size, heatmap_size, sigmas,
                   backend: str = "scipy"):
    """(K,H,W) f32 -> keypoints (1,K,2) float64, scores (1,K) float32.

    ``input_size`` is [w,h], ``heatmap_size`` is [W,H].  Note the asymmetric
    rescale of codec.py:237: divide by (size-1), multiply by input size.
    """
    Wh, Hh = heatmap_size
    locs, vals = heatmap_expected_value(heatmaps.copy(), sigmas, backend=backend)
    kpts = locs[None] / [Wh - 1, Hh - 1] * input_size             # :237 (-> f64)
    return kpts, vals[None]


# --------------------------------------------------------------------------
# C2: Codec.decode, batched                          codec.py:249-263
# --------------------------------------------------------------------------
def codec_decode(pred, input_size, heatmap_size, sigmas, backend: str = "scipy"):
    """5-tuple of arrays/tensors -> ((kpts (B,K,2) f64, scores (B,K) f32),
    prob, vis, oks (B,1,K) f32, err (B,1,K) f64).

    The reference only works for B == 1 (heatmap.py:362-364 raises otherwise);
    batched decode is *defined* as the stack of per-crop B == 1 decodes
    (SURVEY.md section 0).  For B == 1 the shapes are the reference's own:
    kpts (1,K,2), scores (1,K).
    """
    hm, prob, vis, oks, err = [_np(t) for t in pred]
    B, K, H, W = hm.shape
    kp, sc = [], []
    for b in range(B):
        k_, s_ = probmap_decode(hm[b], input_size, heatmap_size, sigmas, backend)
        kp.append(k_[0])
        sc.append(s_[0])
    prob = prob.reshape((B, 1, K))
    vis = vis.reshape((B, 1, K))
    oks = oks.reshape((B, 1, K))
    err = err.reshape((B, 1, K)) / np.sqrt(H ** 2 + W ** 2)       # :260-261 (-> f64)
    return (np.stack(kp), np.stack(sc)), prob, vis, oks, err


def _np(t):
    """util.py:6-12."""
    if isinstance(t, np.ndarray):
        return t
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------
# B1-B6: ViT backbone (timm==1.0.15 semantics; parity unpinned by reference)
#        call sites probpose/backbone.py:26-39
# --------------------------------------------------------------------------
def vit_forward_features(sd: Dict[str, torch.Tensor], x: torch.Tensor, *,
                         patch: int, heads: int, prefix: str = "",
                         explicit_softmax: bool = False) -> torch.Tensor:
    """``VisionTransformer.forward_features`` with class_token=False,
    global_pool='' (=> final ``norm`` applied, no fc_norm), no drop, no
    LayerScale, qkv_bias=True, LN eps 1e-6, exact-erf GELU.  x (B,3,H,W)."""
    g = lambda n: sd[prefix + n]
    depth = 1 + max(int(k[len(prefix) + 7:].split(".")[0]) for k in sd
                    if k.startswith(prefix + "blocks."))
    t = F.conv2d(x, g("patch_embed.proj.weight"), g("patch_embed.proj.bias"), stride=patch)
    t = t.flatten(2).transpose(1, 2)                      # (B,N,C), row-major over (gh,gw)
    t = t + g("pos_embed")
    B, N, C = t.shape
    hd = C // heads
    for i in range(depth):
        p = f"blocks.{i}."
        h = F.layer_norm(t, (C,), g(p + "norm1.weight"), g(p + "norm1.bias"), 1e-6)
        qkv = F.linear(h, g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias"))
        q, k, v = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
        if explicit_softmax:
            a = (q * hd ** -0.5) @ k.transpose(-2, -1)
            o = a.softmax(dim=-1) @ v
        else:
            o = F.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).reshape(B, N, C)
        t = t + F.linear(o, g(p + "attn.proj.weight"), g(p + "attn.proj.bias"))
        h = F.layer_norm(t, (C,), g(p + "norm2.weight"), g(p + "norm2.bias"), 1e-6)
        h = F.gelu(F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")))
        t = t + F.linear(h, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
    return F.layer_norm(t, (C,), g("norm.weight"), g("norm.bias"), 1e-6)


def backbone_forward(sd, x, *, patch: int, heads: int, prefix: str = "model.") -> torch.Tensor:
    """ScratchViTBackbone.forward: (B,N,C) -> (B,C,gh,gw).  backbone.py:35-40."""
    B, _, H, W = x.shape
    f = vit_forward_features(sd, x, patch=patch, heads=heads, prefix=prefix)
    gh, gw = H // patch, W // patch
    return f.reshape(B, gh, gw, -1).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------
# H1-H6: ProbMapHead (eval mode)                     probpose/head.py
# --------------------------------------------------------------------------
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"],
                        sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def sparsemax_lastdim(x: torch.Tensor) -> torch.Tensor:
    """Sparsemax(dim=-1) (head.py:241, third-party ``sparsemax==0.1.9``: NOT in the reference checkout, not
    importable here -> PARITY UNPINNED).  Restated from the published algorithm (Martins & Astudillo 2016,
    "From Softmax to Sparsemax", Algorithm 1) in the order the library's forward evaluates it, in the input's
    dtype: shift by the row maximum, sort descending, cumulative sums, support size
    k = max{ j : 1 + j z_(j) > sum_{i<=j} z_(i) }, tau = (sum_{i<=k} z_(i) - 1) / k, output max(z - tau, 0)."""
    z = x - x.max(dim=-1, keepdim=True).values
    zs = torch.sort(z, dim=-1, descending=True).values
    rng = torch.arange(1, z.shape[-1] + 1, dtype=z.dtype, device=z.device).expand_as(z)
    bound = 1 + rng * zs
    is_gt = (bound > torch.cumsum(zs, dim=-1)).to(z.dtype)
    k = (is_gt * rng).max(dim=-1, keepdim=True).values
    taus = ((is_gt * zs).sum(dim=-1, keepdim=True) - 1) / k
    return torch.max(torch.zeros_like(z), z - taus)


def head_forward_heatmap(sd, x, *, n_deconv: int, final_kernel=1,
                         temperature: float = 0.5, normalize=None, prefix: str = "", conv_kernels: Sequence[int] = ()):
    """head.py:513-534 with the deconv stack head.py:433-474 (k4 s2 p1, no bias), the optional conv stack
    head.py:407-431 (Conv2d k, stride 1, padding (k-1)//2, bias -> BN -> ReLU) and the final layer head.py:227-235
    (Conv2d k, padding k//2; ``final_kernel=None``: nn.Identity)."""
    for i in range(n_deconv):
        x = F.conv_transpose2d(x, sd[f"{prefix}deconv_layers.{3 * i}.weight"], None,
                               stride=2, padding=1, output_padding=0)
        x = F.relu(_bn(sd, f"{prefix}deconv_layers.{3 * i + 1}.", x))
    for i, k in enumerate(conv_kernels):                          # :407-431,524
        x = F.conv2d(x, sd[f"{prefix}conv_layers.{3 * i}.weight"], sd[f"{prefix}conv_layers.{3 * i}.bias"],
                     padding=(k - 1) // 2)
        x = F.relu(_bn(sd, f"{prefix}conv_layers.{3 * i + 1}.", x))
    if final_kernel is not None:
        x = F.conv2d(x, sd[prefix + "final_layer.weight"], sd[prefix + "final_layer.bias"],
                     padding=final_kernel // 2)                   # :227-233,525
    B, C, H, W = x.shape
    x = x.reshape(B, C, H * W) / temperature                      # :527-528
    if normalize is not None:                                     # :528-530 (parity unpinned, see sparsemax_lastdim)
        x = sparsemax_lastdim(x) * normalize
    return torch.clamp(x, 0, 1).reshape(B, C, H, W)               # :531-532


def head_forward_aux(sd, x, name: str, pools: Sequence, last: str, prefix: str = ""):
    """One of probability/visibility/oks/error.  head.py:255-405,536-594:
    [Conv3x3 p1, BN, MaxPool(k,k), ReLU] x n -> Conv1x1 -> Sigmoid | ReLU."""
    n = len(pools)
    for i, k in enumerate(pools):
        p = f"{prefix}{name}_layers."
        x = F.conv2d(x, sd[f"{p}{4 * i}.weight"], sd[f"{p}{4 * i}.bias"], padding=1)
        x = _bn(sd, f"{p}{4 * i + 1}.", x)
        x = F.relu(F.max_pool2d(x, kernel_size=tuple(k) if not isinstance(k, int) else k))
    x = F.conv2d(x, sd[f"{prefix}{name}_layers.{4 * n}.weight"], sd[f"{prefix}{name}_layers.{4 * n}.bias"])
    return torch.sigmoid(x) if last == "sigmoid" else F.relu(x)


def head_forward(sd, feats, *, pools, n_deconv: int = 2, final_kernel=1,
                 prefix: str = "", normalize=None, conv_kernels: Sequence[int] = ()):
    """ProbMapHead.forward (eval).  head.py:487-511."""
    return (
        head_forward_heatmap(sd, feats, n_deconv=n_deconv, final_kernel=final_kernel, prefix=prefix,
                             normalize=normalize, conv_kernels=conv_kernels),
        head_forward_aux(sd, feats, "probability", pools, "sigmoid", prefix),
        head_forward_aux(sd, feats, "visibility", pools, "sigmoid", prefix),
        head_forward_aux(sd, feats, "oks", pools, "sigmoid", prefix),
        head_forward_aux(sd, feats, "error", pools, "relu", prefix),
    )


def model_forward(sd, x, *, patch: int, heads: int, pools, n_deconv: int = 2, normalize=None):
    """ProbPoseModel.forward = head(backbone(x)).  model.py:10-11."""
    feats = backbone_forward(sd, x, patch=patch, heads=heads, prefix="backbone.model.")
    return head_forward(sd, feats, pools=pools, n_deconv=n_deconv, prefix="head.", normalize=normalize)


# --------------------------------------------------------------------------
# Seeded synthetic inputs shared by tests / bench
# --------------------------------------------------------------------------
def synthetic_heatmaps(B: int, K: int, H: int, W: int, seed: int, kind: str = "peaked") -> np.ndarray:
    """Peaked: Gaussian blobs (sigma in [1,3] px, amplitude in [0.3,1]) + 0.02*U
    noise, clamped to [0,1]; every 7th map's peak sits on a border and every
    11th map is all-zero.  Uniform: U[0,1)."""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((B, K, H, W), dtype=np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    out = np.empty((B, K, H, W), np.float32)
    n = 0
    for b in range(B):
        for k in range(K):
            cx, cy = rng.uniform(0, W - 1), rng.uniform(0, H - 1)
            if n % 7 == 3:
                cx = 0.0 if (n // 7) % 2 == 0 else W - 1.0
            sg, amp = rng.uniform(1, 3), rng.uniform(0.3, 1.0)
            m = amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
            m = np.clip(m + 0.02 * rng.random((H, W)), 0, 1)
            if n % 11 == 5:
                m[:] = 0
            out[b, k] = m
            n += 1
    return out


# ---------------------------------------------------------------------------
# Target generation: ProbMap.encode / generate_probmaps
# ---------------------------------------------------------------------------
def generate_probmaps(heatmap_size, keypoints, keypoints_visible, sigmas, sigma=0.55):
    """Restatement of reference codec.py:11-70: per visible keypoint the OKS map
    exp(-dist^2 / (2 s)) over the (H, W) grid, float64 arithmetic, stored float32; s = clip((2 sigma_k)^2 *
    sqrt(H/1.25 * W/1.25) * 2, 0.55, 3.0) unless ``sigma`` > 0 overrides it; weight = (map max > 0)."""
    N, K, _ = keypoints.shape
    W, H = heatmap_size
    heatmaps = np.zeros((K, H, W), dtype=np.float32)
    weights = keypoints_visible.copy()
    bbox_area = np.sqrt(H / 1.25 * W / 1.25)
    yy, xx = np.indices((H, W))
    for n in range(N):
        for k in range(K):
            if keypoints_visible[n, k] < 0.5:
                continue
            dx = xx - keypoints[n, k, 0]
            dy = yy - keypoints[n, k, 1]
            dist = np.sqrt(dx ** 2 + dy ** 2)
            s = np.clip((sigmas[k] * 2) ** 2 * bbox_area * 2, 0.55, 3.0)
            if sigma is not None and sigma > 0:
                s = sigma
            oks = np.exp(-(dist ** 2 / (2 * s)))
            weights[n, k] = (oks.max() > 0).astype(int)
            heatmaps[k] = oks
    return heatmaps, weights


def probmap_encode(keypoints, keypoints_visible, input_size, heatmap_size, sigmas, sigma=2.0):
    """Restatement of reference codec.py:138-212 (the fields the training target uses)."""
    scale_factor = ((np.array(input_size) - 1) / (np.array(heatmap_size) - 1)).astype(np.float32)
    if keypoints_visible is None:
        keypoints_visible = np.ones(keypoints.shape[:2], dtype=np.float32)
    hm_kpts = keypoints / scale_factor
    heatmaps, weights = generate_probmaps(heatmap_size, hm_kpts, keypoints_visible, sigmas, sigma)
    in_image = ((keypoints[:, :, 0] >= 0) & (keypoints[:, :, 0] < input_size[0]) &
                (keypoints[:, :, 1] >= 0) & (keypoints[:, :, 1] < input_size[1]))
    return dict(heatmaps=heatmaps, keypoint_weights=weights, annotated=keypoints_visible > 0, in_image=in_image,
                heatmap_keypoints=hm_kpts)


def synthetic_keypoints(K: int, input_size, seed: int):
    """Seeded (1, K, 2) float32 keypoints in input-image pixels (some outside the image) + (1, K) visibility
    flags in {0, 1, 2}-style floats as dataset.py:116-118 produces them."""
    rng = np.random.default_rng(seed)
    w, h = input_size
    kp = np.stack([rng.uniform(-0.15 * w, 1.15 * w, K), rng.uniform(-0.15 * h, 1.15 * h, K)], -1)[None].astype(np.float32)
    vis = (rng.random((1, K)) > 0.2).astype(np.float32)
    kp[0, 0] = (10.0, 20.0)          # exactly representable, well inside
    kp[0, 1] = (-4000.0, -4000.0)    # so far away that exp underflows: weight 0
    vis[0, 0] = vis[0, 1] = 1.0
    return kp, vis



# --------------------------------------------------------------------------
# f2: ArgMaxProbMap.decode = raw arg-max + DARK-UDP            probpose/codec.py:284-375,515-543
# (cv2.GaussianBlur is third-party and not importable here: PARITY UNPINNED; restated from OpenCV's published
#  getGaussianKernel / separable float32 filtering.  The zero padding of codec.py:305-306 is as wide as the kernel
#  radius, so cv2's own border mode (BORDER_REFLECT_101) never reaches the cropped result.)
# --------------------------------------------------------------------------
def get_heatmap_maximum(heatmaps: np.ndarray):
    """heatmap.py:13-52 (pinned by tests/golden/metrics.npz, minted from the imported reference)."""
    if heatmaps.ndim == 3:
        K, H, W = heatmaps.shape
        B = None
        flat = heatmaps.reshape(K, -1)
    else:
        B, K, H, W = heatmaps.shape
        flat = heatmaps.reshape(B * K, -1)
    y_locs, x_locs = np.unravel_index(np.argmax(flat, axis=1), shape=(H, W))
    locs = np.stack((x_locs, y_locs), axis=-1).astype(np.float32)
    vals = np.amax(flat, axis=1)
    locs[vals <= 0.0] = -1
    if B:
        locs = locs.reshape(B, K, 2)
        vals = vals.reshape(B, K)
    return locs, vals


def gaussian_kernel_f32(ksize: int) -> np.ndarray:
    """cv2.getGaussianKernel(ksize, sigma <= 0, CV_32F) as OpenCV's source computes it."""
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    t = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)
    return (t.astype(np.float64) * (1.0 / t.astype(np.float64).sum())).astype(np.float32)


def gaussian_blur_zero_padded(hm: np.ndarray, ksize: int) -> np.ndarray:
    """codec.py:300-312 for one (H, W) float32 map: zero-pad by the radius, separable float32 Gaussian, crop, rescale to
    the original maximum.  Row pass: taps in index order; column pass: centre tap + symmetric pairs; every product and
    sum rounded to float32 (no fused multiply-add)."""
    k = gaussian_kernel_f32(ksize)
    b = (ksize - 1) // 2
    H, W = hm.shape
    origin_max = np.max(hm)
    P = np.zeros((H + 2 * b, W + 2 * b), dtype=np.float32)
    P[b:b + H, b:b + W] = hm
    R = np.zeros((H + 2 * b, W), dtype=np.float32)
    acc = k[0] * P[b:b + H, 0:W]
    for t in range(1, ksize):
        acc = acc + k[t] * P[b:b + H, t:t + W]
    R[b:b + H] = acc
    out = k[b] * R[b:b + H]
    for t in range(1, b + 1):
        out = out + k[b + t] * (R[b + t:b + t + H] + R[b - t:b - t + H])
    out = out.astype(np.float32)
    out *= origin_max / (np.max(out) + 1e-12)
    return out


def dark_udp_decode(heatmaps: np.ndarray, blur_kernel_size: int, input_size, heatmap_size):
    """ArgMaxProbMap.decode (codec.py:515-543) for one crop's (K, H, W) float32 maps -> (kpts (1,K,2) f64,
    scores (1,K) f32).  A map whose maximum is <= 0 keeps the location (-1, -1) un-refined (the reference reads a
    neighbouring map through negative indices there: defined behaviour instead of that out-of-bounds read)."""
    hm = heatmaps.astype(np.float32).copy()
    W, H = heatmap_size
    K = hm.shape[0]
    locs, scores = get_heatmap_maximum(hm)
    kp = locs.copy()
    for k in range(K):
        if not scores[k] > 0:
            continue
        L = gaussian_blur_zero_padded(hm[k], blur_kernel_size)
        np.clip(L, 1e-3, 50., L)
        np.log(L, L)
        Lp = np.pad(L, ((1, 1), (1, 1)), mode="edge")
        x, y = int(locs[k, 0]) + 1, int(locs[k, 1]) + 1
        i_, ix1, iy1 = Lp[y, x], Lp[y, x + 1], Lp[y + 1, x]
        ix1y1, ix1_y1_, ix1_, iy1_ = Lp[y + 1, x + 1], Lp[y - 1, x - 1], Lp[y, x - 1], Lp[y - 1, x]
        dx = np.float32(0.5) * (ix1 - ix1_)
        dy = np.float32(0.5) * (iy1 - iy1_)
        dxx = ix1 - 2 * i_ + ix1_
        dyy = iy1 - 2 * i_ + iy1_
        dxy = np.float32(0.5) * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_)
        hess = np.array([[dxx, dxy], [dxy, dyy]], dtype=np.float32)
        hinv = np.linalg.pinv(hess + np.finfo(np.float32).eps * np.eye(2))
        kp[k] -= (hinv @ np.array([dx, dy], dtype=np.float32).reshape(2, 1)).squeeze()
    kpts = kp[None] / [W - 1, H - 1] * input_size
    return kpts, scores[None]


# --------------------------------------------------------------------------
# f3 (second half): evaluation metrics, one instance at a time as the reference evaluates them
#   compute_oks            probpose/loss.py:715-764
#   keypoint_pck_accuracy  probpose/loss.py:825-866   (over heatmap.py:55-111)
#   pose_pck_accuracy      probpose/loss.py:767-822   ('argmax' method)
# Pinned bit for bit by tests/golden/metrics.npz (minted from the imported reference).  The product's batched forms
# (probpose_pytorch_amd/metrics.py: oks_batch on arrays, pp_pck_counts on the device) are checked against these and
# against the goldens.
# --------------------------------------------------------------------------
def compute_oks(gt: dict, dt: dict, sigmas: np.ndarray, use_area: bool = True, per_kpt: bool = False):
    """loss.py:715-764: OKS of one detection against one annotation."""
    k = len(sigmas)
    variances = (sigmas * 2) ** 2
    g = np.array(gt["keypoints"]).reshape(k, 3)
    d = np.array(dt["keypoints"]).reshape(k, 3)
    visible = g[:, 2] > 0                                        # loss.py:723-728
    n_visible = np.count_nonzero(visible)
    bx, by, bw, bh = gt["bbox"]
    if n_visible > 0:                                            # loss.py:739-741
        dx, dy = d[:, 0] - g[:, 0], d[:, 1] - g[:, 1]
    else:                                                        # loss.py:742-746: distance to the box grown by its size
        zero = np.zeros(k)
        dx = np.max((zero, (bx - bw) - d[:, 0]), axis=0) + np.max((zero, d[:, 0] - (bx + bw * 2)), axis=0)
        dy = np.max((zero, (by - bh) - d[:, 1]), axis=0) + np.max((zero, d[:, 1] - (by + bh * 2)), axis=0)
    scale = gt["area"] if use_area else bh * bw * 0.53           # loss.py:748-752
    e = (dx ** 2 + dy ** 2) / variances / (scale + np.spacing(1)) / 2
    if per_kpt:                                                  # loss.py:754-757
        oks = np.exp(-e)
        if n_visible > 0:
            oks[~visible] = 0
        return oks
    if n_visible > 0:                                            # loss.py:759-762
        e = e[visible]
    return np.sum(np.exp(-e)) / e.shape[0]


def normalized_distances(preds: np.ndarray, gts: np.ndarray, mask: np.ndarray, norm_factor: np.ndarray) -> np.ndarray:
    """heatmap.py:55-89 -> (K, N) float32, -1 where masked out (``norm_factor`` is edited in place like there)."""
    N, K, _ = preds.shape
    use = mask.copy()
    use[np.where((norm_factor == 0).sum(1))[0], :] = False       # heatmap.py:78-79
    out = np.full((N, K), -1, dtype=np.float32)
    norm_factor[np.where(norm_factor <= 0)] = 1e6                # heatmap.py:82
    out[use] = np.linalg.norm(((preds - gts) / norm_factor[:, None, :])[use], axis=-1)
    return out.T


def keypoint_pck_accuracy(pred, gt, mask, thr, norm_factor):
    """loss.py:825-866 with heatmap.py:92-111 inlined: per-keypoint accuracy (-1: no valid pair), mean, count."""
    acc = []
    for row in normalized_distances(pred, gt, mask, norm_factor):
        valid = row != -1
        acc.append((row[valid] < thr).sum() / valid.sum() if valid.sum() > 0 else -1)
    acc = np.array(acc)
    good = acc[acc >= 0]
    return acc, (good.mean() if len(good) > 0 else 0.0), len(good)


def pose_pck_accuracy(output: np.ndarray, target: np.ndarray, mask: np.ndarray, thr: float = 0.05, normalize=None):
    """loss.py:767-822, method 'argmax'."""
    N, K, H, W = output.shape
    if K == 0:
        return None, 0, 0
    if normalize is None:
        normalize = np.tile(np.array([[H, W]]), (N, 1))
    return keypoint_pck_accuracy(get_heatmap_maximum(output)[0], get_heatmap_maximum(target)[0], mask, thr, normalize)
