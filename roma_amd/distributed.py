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


class PendingGather:
    """Handle of an asynchronous gather_results(): `wait()` returns (warp, certainty) on `dst`, (None, None) elsewhere.

    The collectives run on the backend's own stream; wait() only makes the CURRENT stream wait for them, so a caller
    that first enqueues the next batch's match() and then waits overlaps the transfer with that compute."""

    def __init__(self, works, finish, keep=None):
        self._works, self._finish, self._keep = works, finish, keep  # `keep`: send / receive buffers stay referenced

    def wait(self):
        for w in self._works:
            w.wait()
        out = self._finish()
        self._works, self._keep, self._finish = [], None, (lambda: out)
        return out


def gather_results(warp: torch.Tensor, cert: torch.Tensor, n_pairs: int, dst: int = 0, async_op: bool = False):
    """Gather per-rank (warp [c,H,W,4], certainty [c,H,W]) on `dst` in pair order.

    Equal shards (the usual case: 8 GPUs x 8 pairs, the root receives 7 x 239 MB over 7 independent point-to-point
    xGMI links) land directly in slices of the result tensor - no concatenation pass.  Ragged shards are padded to the
    largest shard so that one gather per tensor still suffices.  With async_op=True a PendingGather is returned."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return PendingGather([], lambda: (warp, cert)) if async_op else (warp, cert)
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_pairs(n_pairs, r, world)[1] for r in range(world)]
    cmax, even = max(counts), len(set(counts)) == 1

    def pad(t):
        if t.shape[0] == cmax:
            return t.contiguous()
        p = torch.zeros((cmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        p[: t.shape[0]] = t
        return p

    works, parts, keep = [], [], []
    for t in (warp, cert):
        src = pad(t)
        buf = full = None
        if rank == dst:
            if even:  # receive straight into the result
                full = torch.empty((n_pairs,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                buf = [full[r * cmax:(r + 1) * cmax] for r in range(world)]
            else:
                buf = [torch.empty_like(src) for _ in range(world)]
        works.append(dist.gather(src, gather_list=buf, dst=dst, async_op=True))
        keep.append((src, buf))
        parts.append((full, buf))

    def finish():
        if rank != dst:
            return None, None
        res = [full if full is not None else torch.cat([buf[r][: counts[r]] for r in range(world)], dim=0)
               for full, buf in parts]
        return res[0], res[1]

    pending = PendingGather(works, finish, keep)
    return pending if async_op else pending.wait()
