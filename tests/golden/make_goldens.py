"""Mint the golden fixtures by running the *reference itself* in this container.

Run once, here (the reference checkout does not travel to the GPU box):

    cd /tmp && python /root/repo/tests/golden/make_goldens.py

What it does
  * registers an empty ``cv2`` stub (cv2 is imported but unused on this path:
    reference heatmap.py:5, codec.py:3) and imports ``probpose`` from
    /root/reference, unmodified;
  * G1  decode_*.npz : per-crop loop of the reference ``Codec.decode`` (it only
    supports B == 1, heatmap.py:362-364) on seeded synthetic heatmaps;
  * G2  convmaps_k17.npz : the intermediate convolved maps of one crop from
    both reference back-ends (``return_heatmap=True``, heatmap.py:392-393);
  * G2b bigmap_256.npz : the shape of the reference's own test
    (tests/test_heatmap.py:6: K=20, 256x256, random sigmas), locs/vals + a
    strided sample of the convolved maps;
  * G3/G4 head_c384.npz : reference ``ProbMapHead(...).eval()`` loaded with the
    seeded synthetic weights of ``probpose_pytorch_amd.synthetic`` on seeded
    features, all 5 outputs, and the reference decode chained on them.
Inputs are regenerated from seeds by the tests; only expected outputs (and
input checksums) are stored.
"""
import hashlib
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

import importlib.util  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# our own packages are loaded by path so that ``import probpose`` below resolves
# to the reference checkout and nothing else
orc = _load("pp_oracle_for_goldens", os.path.join(REPO, "oracle", "probpose_oracle.py"))

sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, REF)
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
import probpose  # noqa: E402

assert all(os.path.abspath(p).startswith(REF) for p in probpose.__path__), list(probpose.__path__)
from probpose.codec import Codec, ProbMap  # noqa: E402
from probpose.head import ProbMapHead  # noqa: E402
from probpose.heatmap import get_heatmap_expected_value  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_decode_batch(codec, pred_np):
    """Reference Codec.decode, one crop at a time, stacked."""
    B = pred_np[0].shape[0]
    outs = []
    for b in range(B):
        one = tuple(torch.from_numpy(p[b:b + 1].copy()) for p in pred_np)
        outs.append(codec.decode(one))
    kpts = np.concatenate([o[0][0] for o in outs], 0)
    scores = np.concatenate([o[0][1] for o in outs], 0)
    rest = [np.concatenate([o[i] for o in outs], 0) for i in range(1, 5)]
    return kpts, scores, rest


def k133_sigmas():
    return np.random.default_rng(133).uniform(0.02, 0.11, 133)


def decode_fixture(name, B, K, H, W, in_size, sigmas, seed, kind):
    hm = orc.synthetic_heatmaps(B, K, H, W, seed, kind)
    rng = np.random.default_rng(seed + 1000)
    aux = [rng.random((B, K, 1, 1), dtype=np.float32) for _ in range(4)]
    codec = Codec(ProbMap(in_size, (W, H), sigmas))
    kpts, scores, rest = ref_decode_batch(codec, (hm, *aux))
    np.savez_compressed(
        os.path.join(HERE, name), B=B, K=K, H=H, W=W, in_size=np.array(in_size), sigmas=sigmas,
        seed=seed, kind=kind, hm_sha=sha(hm), kpts=kpts, scores=scores, prob=rest[0], vis=rest[1],
        oks=rest[2], err=rest[3])
    print(name, kpts.shape, kpts.dtype, scores.dtype, [r.dtype for r in rest])


def main():
    coco = orc.COCO17_SIGMAS
    decode_fixture("decode_k17_peaked.npz", 8, 17, 64, 48, (192, 256), coco, 4321, "peaked")
    decode_fixture("decode_k17_uniform.npz", 2, 17, 64, 48, (192, 256), coco, 99, "uniform")
    decode_fixture("decode_k133_peaked.npz", 2, 133, 96, 72, (288, 384), k133_sigmas(), 777, "peaked")
    decode_fixture("decode_k20_sq96.npz", 2, 20, 96, 96, (384, 384), np.array([0.05] * 20), 5, "peaked")

    # G2: intermediate convolved maps, both back-ends
    hm = orc.synthetic_heatmaps(1, 17, 64, 48, 4321, "peaked")[0]
    l_s, v_s, c_s = get_heatmap_expected_value(hm, coco, return_heatmap=True, backend="scipy")
    l_t, v_t, c_t = get_heatmap_expected_value(hm, coco, return_heatmap=True, backend="torch")
    np.savez_compressed(os.path.join(HERE, "convmaps_k17.npz"), hm_sha=sha(hm), conv_scipy=c_s,
                        conv_torch=c_t, locs=l_s, vals=v_s, locs_torch=l_t)
    print("convmaps max|scipy-torch|", np.abs(c_s - c_t).max())

    # G2b: the reference test's own shape (tests/test_heatmap.py:6-7), seeded here
    rng = np.random.default_rng(2024)
    big = rng.random((20, 256, 256), dtype=np.float32)
    sig = rng.random(20, dtype=np.float32)
    l_b, v_b, c_b = get_heatmap_expected_value(big, sig, return_heatmap=True, backend="scipy")
    np.savez_compressed(os.path.join(HERE, "bigmap_256.npz"), hm_sha=sha(big), sigmas=sig, locs=l_b,
                        vals=v_b, conv_sample=c_b[:, ::16, ::16].copy())
    print("bigmap", l_b.shape)

    # G6: target generation, reference ProbMap.encode (codec.py:138-212) on seeded keypoints
    for name, K, in_size, hm_size, sigmas, sigma, seed in (
            ("encode_k17_sigma2", 17, (192, 256), (48, 64), coco, 2.0, 31),
            ("encode_k17_persigma", 17, (192, 256), (48, 64), coco, None, 32),
            ("encode_k133_sigma2", 133, (288, 384), (72, 96), k133_sigmas(), 2.0, 33)):
        kp, vis = orc.synthetic_keypoints(K, in_size, seed)
        enc = ProbMap(in_size, hm_size, sigmas, sigma=sigma).encode(kp.copy(), vis.copy())
        hm = enc["heatmaps"]
        keep = hm if K == 17 else hm[:12]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kp_sha=sha(kp), heatmaps=keep, hm_sha=sha(hm),
                            weights=enc["keypoint_weights"], annotated=enc["annotated"], in_image=enc["in_image"],
                            heatmap_keypoints=enc["heatmap_keypoints"], sigma=np.array(-1.0 if sigma is None else sigma))
        print(name, hm.shape, hm.dtype, "weights", enc["keypoint_weights"].ravel()[:4])

    # G3 / G4: head
    try:
        syn = _load("pp_synthetic_for_goldens", os.path.join(REPO, "probpose_pytorch_amd", "synthetic.py"))
    except FileNotFoundError:
        print("synthetic.py not present yet: head goldens skipped")
        return
    C, K, pools = 384, 17, [(4, 3), (2, 2), (2, 2)]
    head = ProbMapHead(C, K, pools, (256, 256), (4, 4), final_layer_kernel_size=1).eval()
    sd = syn.synthetic_head_state(C, K, n_pools=len(pools), deconv_out=(256, 256), seed=11)
    missing = head.load_state_dict(sd, strict=True)
    print("head load:", missing)
    feats = syn.synthetic_features(2, C, 16, 12, seed=12)
    with torch.no_grad():
        out = head(feats)
    out_np = [o.numpy() for o in out]
    codec = Codec(ProbMap((192, 256), (48, 64), coco))
    kpts, scores, rest = ref_decode_batch(codec, out_np)
    np.savez_compressed(
        os.path.join(HERE, "head_c384.npz"), C=C, K=K, pools=np.array(pools), feats_sha=sha(feats.numpy()),
        w_probe=sd["final_layer.weight"].numpy().ravel()[:8],
        heatmaps=out_np[0].astype(np.float16 if False else np.float32),
        prob=out_np[1], vis=out_np[2], oks=out_np[3], err=out_np[4], kpts=kpts, scores=scores,
        dec_err=rest[3])
    print("head", [o.shape for o in out_np], "hm range", out_np[0].min(), out_np[0].max(),
          "frac>0", (out_np[0] > 0).mean())


if __name__ == "__main__":
    main()
