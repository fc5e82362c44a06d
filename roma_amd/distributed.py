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


def symmetric_grid(H: int, W: int, device, dtype=torch.float32) -> torch.Tensor:
    """[H, W, 2] (x, y): the im_A / im_B coordinate grid every symmetric match() result carries in warp[:, :, :W, :2] and
    warp[:, :, W:, 2:] (romatch/models/matcher.py:904-924).  torch.linspace evaluated on the CPU, as the reference's goldens
    were - the HIP epilogue reproduces those bits (tests: grid_channels_exact)."""
    ys = torch.linspace(-1 + 1 / H, 1 - 1 / H, H, dtype=dtype)
    xs = torch.linspace(-1 + 1 / W, 1 - 1 / W, W, dtype=dtype)
    gy, gx = torch.meshgrid((ys, xs), indexing="ij")
    return torch.stack((gx, gy), dim=-1).to(device)


def compact_symmetric_warp(warp: torch.Tensor) -> torch.Tensor:
    """[c, H, 2W, 4] -> [c, H, 2W, 2]: the two PREDICTED channels of each half (left: [..., 2:], right: [..., :2])."""
    W = warp.shape[2] // 2
    return torch.cat((warp[:, :, :W, 2:], warp[:, :, W:, :2]), dim=2).contiguous()


def expand_symmetric_warp(pred: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """Inverse of compact_symmetric_warp given the [H, W, 2] grid."""
    c, H, W2, _ = pred.shape
    W = W2 // 2
    out = torch.empty((c, H, W2, 4), dtype=pred.dtype, device=pred.device)
    out[:, :, :W, :2] = grid
    out[:, :, :W, 2:] = pred[:, :, :W]
    out[:, :, W:, :2] = pred[:, :, W:]
    out[:, :, W:, 2:] = grid
    return out


def gather_results(warp: torch.Tensor, cert: torch.Tensor, n_pairs: int, dst: int = 0, async_op: bool = False,
                   compact_grid: bool = False):
    """Gather per-rank (warp [c,H,W,4], certainty [c,H,W]) on `dst` in pair order.

    Equal shards (the usual case: 8 GPUs x 8 pairs, the root receives 7 x 239 MB over 7 independent point-to-point
    xGMI links) land directly in slices of the result tensor - no concatenation pass.  Ragged shards are padded to the
    largest shard so that one gather per tensor still suffices.  With async_op=True a PendingGather is returned.

    compact_grid=True (symmetric results only: warp [c,H,2W,4]): half of every warp is the constant coordinate grid
    (matcher.py:904-924), so only the two predicted channels of each half travel (2 of 4 floats per pixel; with the
    certainty 3 of 5: -40 % of the root's ingress, 7 x 143 MB instead of 7 x 239 MB) and the root rebuilds the grid
    channels - from its own shard's result when it has one (the same bits by construction), else from the reference's
    linspace formula.  The gathered tensors are byte-identical to the plain gather's
    (tests/test_cpu_oracle.py::test_gloo_world2_gather)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return PendingGather([], lambda: (warp, cert)) if async_op else (warp, cert)
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_pairs(n_pairs, r, world)[1] for r in range(world)]
    cmax, even = max(counts), len(set(counts)) == 1
    grid = None
    if compact_grid:
        assert warp.dim() == 4 and warp.shape[-1] == 4 and warp.shape[2] % 2 == 0, "compact_grid needs symmetric warps [c,H,2W,4]"
        H, W = warp.shape[1], warp.shape[2] // 2
        if rank == dst:
            grid = warp[0, :, :W, :2].clone() if warp.shape[0] > 0 else symmetric_grid(H, W, warp.device, warp.dtype)
        warp = compact_symmetric_warp(warp)

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
        if grid is not None:
            res[0] = expand_symmetric_warp(res[0], grid)
        return res[0], res[1]

    pending = PendingGather(works, finish, keep)
    return pending if async_op else pending.wait()
