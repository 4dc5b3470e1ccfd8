"""Data-parallel sharding of person crops + the one collective of the path.

The reference has no multi-device code; nothing on the
path mixes batch elements in eval mode, so crops shard contiguously over the
ranks (one process per GPU), weights are replicated, and the only exchange is
a single all-gather of the *decoded* results: per crop K x 7 numbers
(x, y, score, prob, vis, oks, err) instead of K x H x W heatmaps.  On ROCm the
"nccl" backend is RCCL; over the fully connected xGMI mesh this ~60 KB/rank
message is latency-bound, so it is issued once per batch on the compute stream.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

FIELDS = ("x", "y", "score", "prob", "vis", "oks", "err")

# receive / padding buffers of the all-gather, keyed by (shape, dtype, device): allocated once, reused every step
# (no allocation inside the step; the addresses stay stable for graph capture around the collective)
_BUFFERS: Dict[tuple, torch.Tensor] = {}


def _buffer(tag: str, shape, dtype, device) -> torch.Tensor:
    key = (tag, tuple(shape), dtype, str(device))
    t = _BUFFERS.get(key)
    if t is None:
        if len(_BUFFERS) >= 16:
            _BUFFERS.pop(next(iter(_BUFFERS)))
        t = torch.zeros(shape, dtype=dtype, device=device)
        _BUFFERS[key] = t
    return t


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank's crops; the first total % world ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_decoded(out: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Decode outputs (Codec.decode_device) -> one float64 tensor [B, K, 7] (lossless for the f32 fields)."""
    if "packed" in out:          # written by the decode kernel itself
        return out["packed"]
    kpts = out["kpts"]
    B, K, _ = kpts.shape
    packed = torch.empty((B, K, 7), dtype=torch.float64, device=kpts.device)
    packed[..., 0:2] = kpts
    packed[..., 2] = out["scores"]
    packed[..., 3:6] = out["aux"].permute(1, 2, 0)
    packed[..., 6] = out["err"]
    return packed


def unpack_decoded(packed: torch.Tensor):
    """[B,K,7] f64 -> the structure Codec.decode returns (numpy, reference dtypes)."""
    p = packed.cpu().numpy()
    B, K, _ = p.shape
    f32 = lambda a: a.astype("float32").reshape(B, 1, K)
    return ((p[..., 0:2].copy(), p[..., 2].astype("float32")), f32(p[..., 3]), f32(p[..., 4]), f32(p[..., 5]),
            p[..., 6].reshape(B, 1, K).copy())


def all_gather_decoded(packed: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                       sizes: Optional[Tuple[int, ...]] = None) -> torch.Tensor:
    """All-gather the per-rank [B_r, K, 7] blocks into [sum B_r, K, 7] (rank order = crop order).

    ``sizes`` gives every rank's B_r when shards are uneven (they are padded to the
    largest shard for the collective and trimmed afterwards)."""
    if not dist.is_available() or not dist.is_initialized():
        return packed
    world = dist.get_world_size(group)
    if world == 1:
        return packed
    Bmax = max(sizes) if sizes is not None else packed.shape[0]
    send = packed
    if packed.shape[0] != Bmax:
        send = _buffer("send", (Bmax,) + tuple(packed.shape[1:]), packed.dtype, packed.device)
        send[: packed.shape[0]] = packed
        send[packed.shape[0]:] = 0
    send = send.contiguous()
    if dist.get_backend(group) == "nccl":
        recv = _buffer("recv", (world * Bmax,) + tuple(send.shape[1:]), send.dtype, send.device)
        dist.all_gather_into_tensor(recv, send, group=group)
        if sizes is None:
            return recv                      # even shards: the gathered buffer already is [world * B, K, 7]
                                             # (reused by the next call: consume or copy it before gathering again)
        parts = list(recv.split(Bmax))
    else:
        # gloo (CPU tests, single-GPU rehearsals of the multi-rank path): the collective runs on host copies
        host = send.cpu() if send.is_cuda else send
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        if send.is_cuda:
            parts = [p.to(send.device) for p in parts]
    if sizes is not None:
        parts = [p[:n] for p, n in zip(parts, sizes)]
    return torch.cat(parts, 0)
