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


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher around it (the way the driver runs --gpus 1) must measure N GPUs:
    the parent assembles the torch.distributed.run command (rendezvous on 127.0.0.1), runs it as a CHILD (never an exec),
    passes its own arguments through and returns the children's status; under a launcher (WORLD_SIZE set) it does not."""
    import subprocess
    import sys

    import bench
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "7", "--warmup", "2"], 4, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(bench.__file__ if bench.__file__ in cmd else next(c for c in cmd if c.endswith("bench.py")))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    calls = []

    def fake_call(argv, **kw):
        calls.append((argv, kw))
        return 0 if len(calls) == 1 else 3          # build ok, ranks exit 3

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.launch_ranks(["--gpus", "2"], 2) == 3
    assert len(calls) == 2 and "g.build()" in calls[0][0][-1]          # the build runs in a child, before the ranks
    assert "--nproc-per-node=2" in calls[1][0] and calls[1][1]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # main(): --gpus 2 with no WORLD_SIZE goes through launch_ranks and exits with its status ...
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    monkeypatch.setattr(bench, "launch_ranks", lambda argv, n: 17 if (argv, n) == (["--gpus", "2", "--steps", "1"], 2) else 1)
    try:
        bench.main()
        raise AssertionError("main() should have exited")
    except SystemExit as e:
        assert e.code == 17


@pytest.mark.gpu
def test_bench_gpus2_launches_two_ranks_on_the_gpu_box(tmp_path):
    """`python bench.py --gpus 2` started the way the driver starts `--gpus 1` (no launcher, no WORLD_SIZE): the parent
    spawns its own two ranks and relays rank 0's line with n_gpus = 2, parallelism dp2, global batch 2 x 64.  On the
    one-GPU box both ranks share the card and the collective runs over gloo (PP_BENCH_REHEARSAL=1: RCCL refuses two
    ranks on one device), so this walks the launcher, the rendezvous, the sharded seeds, the all-gather call path and
    the max-over-ranks timing -- everything but RCCL itself."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PP_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-parity", "--no-decode-scale"], cwd=root, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 128
    assert d["scaling"] == "weak" and d["value"] > 0
