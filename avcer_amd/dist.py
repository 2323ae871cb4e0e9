"""Clip sharding across the GPUs of one node and the single collective of the path.

Clips are independent: rank r processes the contiguous block [r*N/W, (r+1)*N/W) with the
weights replicated, and the only exchange is ONE all-gather of the per-clip record
{static_probs[T,7], dyn_logits[T,7], audio_logits[C]} (928 B per clip at T = 16 frames and 8 audio classes) before the fusion, which then runs
replicated.  backend "nccl" is RCCL over xGMI on ROCm; the same code runs on gloo/CPU tensors in the tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int):
    """Contiguous block partition; the first n % world ranks get one extra clip."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_records(stat: torch.Tensor, dyn: torch.Tensor, aud: torch.Tensor) -> torch.Tensor:
    n = stat.shape[0]
    return torch.cat([stat.reshape(n, -1), dyn.reshape(n, -1), aud.reshape(n, -1)], dim=1).contiguous()


def unpack_records(rec: torch.Tensor, t: int, c: int):
    n = rec.shape[0]
    stat = rec[:, :t * 7].reshape(n, t, 7)
    dyn = rec[:, t * 7:2 * t * 7].reshape(n, t, 7)
    aud = rec[:, 2 * t * 7:2 * t * 7 + c].reshape(n, c)
    return stat, dyn, aud


def all_gather_records(rec: torch.Tensor, n_total: int, force: bool = False) -> torch.Tensor:
    """All-gather of row blocks produced with shard_range (uneven blocks are padded to the largest one).
    `force`: run the collective at world size 1 as well (tests/test_gpu_rccl.py: the RCCL path on a one-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return rec
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    pad = torch.zeros(mx, rec.shape[1], dtype=rec.dtype, device=rec.device)
    pad[:rec.shape[0]] = rec
    out = torch.empty(world * mx, rec.shape[1], dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, pad)
    parts = [out[r * mx:r * mx + (b - a)] for r, (a, b) in enumerate(sizes)]
    return torch.cat(parts, dim=0)
