"""CPU tests of the data-parallel path: shard bounds + the all-gather of
decoded results over gloo with world_size 2 (one process per rank)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from probpose_pytorch_amd import parallel


def test_shard_bounds_cover_the_batch_exactly():
    for total in (0, 1, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_bounds(8, 2, 2)


def _fake_decoded(B, K, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(kpts=torch.rand((B, K, 2), generator=g, dtype=torch.float64) * 200,
                scores=torch.rand((B, K), generator=g), aux=torch.rand((3, B, K), generator=g),
                err=torch.rand((B, K), generator=g, dtype=torch.float64))


def test_pack_unpack_roundtrip_is_lossless():
    out = _fake_decoded(5, 17, 0)
    (kpts, scores), prob, vis, oks, err = parallel.unpack_decoded(parallel.pack_decoded(out))
    np.testing.assert_array_equal(kpts, out["kpts"].numpy())
    np.testing.assert_array_equal(scores, out["scores"].numpy())
    np.testing.assert_array_equal(prob[:, 0], out["aux"][0].numpy())
    np.testing.assert_array_equal(oks[:, 0], out["aux"][2].numpy())
    np.testing.assert_array_equal(err[:, 0], out["err"].numpy())
    assert kpts.dtype == np.float64 and scores.dtype == np.float32 and err.dtype == np.float64


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = parallel.pack_decoded(_fake_decoded(total, 17, 123))   # every rank can rebuild the whole batch
        sizes = tuple(hi - lo for lo, hi in (parallel.shard_bounds(total, world, r) for r in range(world)))
        lo, hi = parallel.shard_bounds(total, world, rank)
        gathered = parallel.all_gather_decoded(full[lo:hi].clone(), sizes=sizes)
        q.put((rank, bool(torch.equal(gathered, full)), tuple(gathered.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_all_gather_decoded_gloo_world2(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2] == (total, 17, 7) for r in res)
