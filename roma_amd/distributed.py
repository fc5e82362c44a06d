"""Multi-GPU use of the match() path: one process per GPU, image pairs sharded contiguously.

The path is embarrassingly parallel over pairs (no cross-pair dependence anywhere in match():
BatchNorm is in eval mode, GP / attention are per sample), so there is NO data-path collective;
the only exchange is one gather of the results to the root (RCCL over xGMI on GPUs, gloo on CPU
in tests).  The reference has no multi-GPU inference path (its only collectives are DDP gradient
all-reduces in experiments/train_roma_outdoor.py:169-251).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_pairs(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: returns (start, count) of this rank's pairs."""
    base, rem = divmod(n_pairs, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def gather_results(warp: torch.Tensor, cert: torch.Tensor, n_pairs: int, dst: int = 0):
    """Gather per-rank (warp [c,H,W,4], certainty [c,H,W]) on `dst` in pair order.

    Ragged shards are padded to the largest shard so that one gather per tensor suffices
    (8 GPUs x 8 pairs: the root receives 7 x 239 MB over 7 independent point-to-point xGMI links)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return warp, cert
    world, rank = dist.get_world_size(), dist.get_rank()
    cmax = max(shard_pairs(n_pairs, r, world)[1] for r in range(world))

    def pad(t):
        if t.shape[0] == cmax:
            return t.contiguous()
        p = torch.zeros((cmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        p[: t.shape[0]] = t
        return p

    outs = []
    for t in (warp, cert):
        buf = None
        if rank == dst:
            buf = [torch.empty((cmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for _ in range(world)]
        dist.gather(pad(t), gather_list=buf, dst=dst)
        if rank == dst:
            outs.append(torch.cat([buf[r][: shard_pairs(n_pairs, r, world)[1]] for r in range(world)], dim=0))
        else:
            outs.append(None)
    return outs[0], outs[1]
