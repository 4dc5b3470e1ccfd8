"""Mint tests/golden/head_variants.npz: outputs of the REFERENCE ProbMapHead (imported from /root/reference, eval mode)
for the constructor branches the main head golden does not touch:
  A  conv_out_channels / conv_kernel_sizes (head.py:407-431) + final_layer_kernel_size = 3 (head.py:227-233)
  B  a conv stack ending in K channels + final_layer_kernel_size = None -> nn.Identity (head.py:234-235)
  C  deconv_out_channels = () (nn.Identity deconv stack, head.py:203-204) + a 1x1 conv stack + 1x1 final layer
cv2 is registered as an empty stub exactly as in make_goldens.py (imported by the reference package, unused here).
Run once, here:  cd /tmp && python /root/repo/tests/golden/make_goldens_head_variants.py
Only seeds and outputs are stored; the tests regenerate weights and inputs from the seeds."""
import importlib.util
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, REF)
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
import probpose  # noqa: E402

assert all(os.path.abspath(p).startswith(REF) for p in probpose.__path__)
from probpose.head import ProbMapHead  # noqa: E402

VARIANTS = {
    # name: (C, K, pools, feature hw, deconv_out, conv_out, conv_kernels, final_kernel, seed)
    "A": (128, 17, [(4, 3), (2, 2), (2, 2)], (16, 12), (64, 64), (64,), (3,), 3, 21),
    "B": (128, 17, [(4, 3), (2, 2), (2, 2)], (16, 12), (64,), (64, 17), (3, 1), None, 22),
    "C": (64, 5, [(4, 3), (2, 2)], (8, 6), (), (64,), (1,), 1, 23),
}


def main():
    spec = importlib.util.spec_from_file_location("pp_synthetic_for_goldens",
                                                  os.path.join(REPO, "probpose_pytorch_amd", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    out = {}
    for name, (C, K, pools, (h, w), dec, conv, ck, fk, seed) in VARIANTS.items():
        head = ProbMapHead(C, K, pools, dec, (4,) * len(dec), conv_out_channels=conv or None,
                           conv_kernel_sizes=ck or None, final_layer_kernel_size=fk).eval()
        sd = syn.synthetic_head_state(C, K, n_pools=len(pools), deconv_out=dec, seed=seed, final_kernel=fk,
                                      conv_out=conv, conv_kernels=ck)
        print(name, "load:", head.load_state_dict(sd, strict=True))
        feats = syn.synthetic_features(2, C, h, w, seed=seed + 100)
        with torch.no_grad():
            res = head(feats)
        for key, t in zip(("heatmaps", "prob", "vis", "oks", "err"), res):
            out[f"{name}_{key}"] = t.numpy()
        print(name, [tuple(t.shape) for t in res], "hm range", float(res[0].min()), float(res[0].max()),
              "frac in (0,1)", float(((res[0] > 0) & (res[0] < 1)).float().mean()))
    np.savez_compressed(os.path.join(HERE, "head_variants.npz"), **out)


if __name__ == "__main__":
    main()
