"""Sparsemax normalisation of the heatmap rows (reference head.py:237-245, 526-532).

The reference takes the operator from the third-party ``sparsemax==0.1.9`` package, which is neither part of the
reference checkout nor installed here: PARITY UNPINNED.  What is checked instead:
  * CPU: the oracle's restatement (published algorithm, Martins & Astudillo 2016) against the defining properties
    of the projection onto the simplex -- non-negativity, sum 1, the support rule / KKT conditions, closed forms --
    and against an independent float64 evaluation;
  * GPU: ``pp_sparsemax_rows`` against that restatement (<= 1e-6) on the BASELINE row lengths, ragged lengths,
    rows larger than LDS, adversarial rows for the Michelot iteration, and the head end to end with
    ``ProbMapHead(..., normalize=1.0)`` (the only constructor call the reference makes, train.py:44).
"""
import numpy as np
import pytest
import torch

from oracle import probpose_oracle as orc


def _exact64(z: np.ndarray) -> np.ndarray:
    """Independent float64 projection: tau by bisection on sum(max(z - tau, 0)) = 1, then the closed form."""
    z = z.astype(np.float64)
    out = np.empty_like(z)
    for r in range(z.shape[0]):
        lo, hi = z[r].max() - 1.0, z[r].max()
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if np.maximum(z[r] - mid, 0).sum() > 1.0:
                lo = mid
            else:
                hi = mid
        supp = z[r] > lo
        tau = (z[r][supp].sum() - 1.0) / supp.sum()
        out[r] = np.maximum(z[r] - tau, 0)
    return out


def _rows(n, rows, seed, scale):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((rows, n), generator=g) * scale


@pytest.mark.parametrize("n,scale", [(7, 1.0), (3072, 0.2), (3072, 3.0), (6912, 1.0), (1000, 10.0)])
def test_oracle_sparsemax_is_the_simplex_projection(n, scale):
    z = _rows(n, 6, n, scale)
    p = orc.sparsemax_lastdim(z)
    assert p.dtype == torch.float32 and bool((p >= 0).all())
    torch.testing.assert_close(p.sum(-1), torch.ones(6), rtol=0, atol=2e-5)
    # KKT / support rule: on the support p_i = z_i - tau with one tau per row; off the support z_i <= tau
    zs = z - z.max(-1, keepdim=True).values
    for r in range(6):
        supp = p[r] > 0
        tau = (zs[r][supp] - p[r][supp]).double()
        assert float(tau.max() - tau.min()) < 1e-5
        assert float(zs[r][~supp].max() if (~supp).any() else -np.inf) <= float(tau.mean()) + 1e-6
    np.testing.assert_allclose(p.numpy(), _exact64(z.numpy()), rtol=0, atol=1e-6)


def test_oracle_sparsemax_closed_forms():
    # two coordinates: p = clip((a - b + 1) / 2, 0, 1); uniform rows; a row with one dominant entry is one-hot
    z = torch.tensor([[0.3, -0.1], [2.0, 0.0], [-1.0, -1.0]])
    p = orc.sparsemax_lastdim(z)
    torch.testing.assert_close(p, torch.tensor([[0.7, 0.3], [1.0, 0.0], [0.5, 0.5]]), rtol=0, atol=1e-6)
    z = torch.full((2, 3072), 0.25)
    torch.testing.assert_close(orc.sparsemax_lastdim(z), torch.full((2, 3072), 1.0 / 3072), rtol=0, atol=1e-9)
    z = torch.zeros((1, 100))
    z[0, 17] = 1.5
    p = orc.sparsemax_lastdim(z)
    assert float(p[0, 17]) == 1.0 and float(p.sum()) == 1.0


def test_oracle_head_normalize_path():
    """head.py:526-532 with normalize: every map sums to `normalize` (before the clamp binds) and lies in [0, 1]."""
    from probpose_pytorch_amd.synthetic import synthetic_model_state
    img, C, K, pools = (64, 48), 64, 5, [(4, 3)]
    sd = synthetic_model_state(img, 16, C, 1, K, len(pools), (32, 32), seed=1)
    sd = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
    feats = torch.randn((2, C, 4, 3), generator=torch.Generator().manual_seed(3))
    hm = orc.head_forward(sd, feats, pools=pools, normalize=1.0)[0]
    assert hm.shape == (2, K, 16, 12) and float(hm.min()) >= 0 and float(hm.max()) <= 1
    torch.testing.assert_close(hm.sum((-1, -2)), torch.ones(2, K), rtol=0, atol=1e-5)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n,rows,scale", [(3072, 1088, 1.0), (6912, 266, 0.5), (9216, 40, 2.0), (1000, 33, 4.0),
                                          (7, 5, 1.0), (50000, 3, 1.0)])
def test_sparsemax_kernel_matches_restatement(built_lib, n, rows, scale):
    from probpose_pytorch_amd import ops
    z = _rows(n, rows, 100 + n, scale)
    want = orc.sparsemax_lastdim(z)
    got = ops.sparsemax_rows(z.cuda().contiguous(), 1.0).cpu()
    torch.testing.assert_close(got, want, rtol=0, atol=1e-6)
    # * normalize and clamp (head.py:529-531)
    got2 = ops.sparsemax_rows(z.cuda().contiguous(), 40.0).cpu()
    torch.testing.assert_close(got2, torch.clamp(want * 40.0, 0, 1), rtol=0, atol=4e-5)


@pytest.mark.gpu
def test_sparsemax_kernel_edge_rows(built_lib):
    from probpose_pytorch_amd import ops
    n = 3072
    rows = []
    rows.append(torch.full((n,), 0.3))                                   # uniform: 1 / n everywhere
    one = torch.zeros(n); one[5] = 7.0; rows.append(one)                 # one-hot
    rows.append(-torch.arange(n, dtype=torch.float32) * 1e-3)           # slow ramp: large support
    rows.append(-(1.0 - 0.5 ** torch.arange(n, dtype=torch.float32)))   # geometric: one element leaves per Michelot step
    rows.append(-torch.log1p(torch.arange(n, dtype=torch.float32)))     # harmonic-like decay
    rows.append(torch.cat([torch.zeros(n // 2), torch.full((n - n // 2,), -1e-7)]))   # near-ties around tau
    z = torch.stack(rows)
    want = torch.from_numpy(_exact64(z.numpy())).float()
    got = ops.sparsemax_rows(z.cuda().contiguous(), 1.0).cpu()
    torch.testing.assert_close(got, want, rtol=0, atol=1e-6)
    torch.testing.assert_close(got.sum(-1), torch.ones(len(rows)), rtol=0, atol=2e-5)
    assert ops.sparsemax_rows(torch.empty((0, 16), device="cuda"), 1.0).shape == (0, 16)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_head_with_normalize_matches_oracle(built_lib, dtype):
    """ProbMapHead(..., normalize=1.0) (reference train.py:44) on the HIP path vs the oracle head."""
    from probpose_pytorch_amd.head import ProbMapHead, Sparsemax
    from probpose_pytorch_amd.synthetic import synthetic_model_state
    img, C, K, pools = (256, 192), 128, 17, [(4, 3), (2, 2), (2, 2)]
    head = ProbMapHead(C, K, pools, (256, 256), (4, 4), final_layer_kernel_size=1, normalize=1.0)
    assert isinstance(head.normalize_layer, Sparsemax)
    sd = synthetic_model_state(img, 16, C, 1, K, len(pools), (256, 256), seed=2)
    sd = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
    head.load_state_dict(sd)
    head = head.cuda().eval().set_compute_dtype(dtype)
    feats = torch.randn((3, C, 16, 12), generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        got = head(feats.cuda())
        want = orc.head_forward(sd, feats, pools=pools, normalize=1.0)
    hm = got[0].cpu()
    assert hm.shape == (3, K, 64, 48)
    torch.testing.assert_close(hm.sum((-1, -2)), torch.ones(3, K), rtol=0, atol=1e-4)
    if dtype == torch.float32:
        torch.testing.assert_close(hm, want[0], rtol=0, atol=1e-5)
        for g, w in zip(got[1:], want[1:]):
            torch.testing.assert_close(g.cpu(), w, rtol=0, atol=1e-4)
    else:
        assert float((hm - want[0]).abs().max()) < 0.05
    # the stand-alone module (what the reference's normalize_layer is) runs the same kernel
    z = torch.randn((2, 5, 300), generator=torch.Generator().manual_seed(6))
    torch.testing.assert_close(Sparsemax()(z.cuda()).cpu(), orc.sparsemax_lastdim(z), rtol=0, atol=1e-6)
