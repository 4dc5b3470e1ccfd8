"""Shared helpers for the test-suite (inputs are regenerated from seeds and
checked against the sha256 stored in the golden fixtures)."""
import hashlib
import os

import numpy as np

from oracle import probpose_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_decode_fixture(name):
    g = np.load(os.path.join(GOLDEN, name))
    B, K, H, W = int(g["B"]), int(g["K"]), int(g["H"]), int(g["W"])
    seed, kind = int(g["seed"]), str(g["kind"])
    hm = orc.synthetic_heatmaps(B, K, H, W, seed, kind)
    assert sha(hm) == str(g["hm_sha"]), "seeded input drifted from the one the golden was minted on"
    rng = np.random.default_rng(seed + 1000)
    aux = [rng.random((B, K, 1, 1), dtype=np.float32) for _ in range(4)]
    return g, hm, aux


DECODE_FIXTURES = ["decode_k17_peaked.npz", "decode_k17_uniform.npz", "decode_k133_peaked.npz",
                   "decode_k20_sq96.npz"]
