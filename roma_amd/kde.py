"""Gaussian kernel density of sampled matches on HIP - mirror of `romatch/utils/kde.py:4-12`.

    kde(x, std=0.1, half=True, down=None) -> density [n]

The reference builds the full n x n fp16 distance matrix with `torch.cdist` (3.2 GB at the 40 000 samples that
`RegressionMatcher.sample` draws); `roma_op_kde` evaluates the same sum in one all-pairs kernel without the matrix.
`half=True` rounds the coordinates to fp16 like the reference's `x.half()`; the squared distances, the exponential
and the row sum are f32, and the result is returned as f32 (the reference returns fp16 when `half=True`).
torch is only the tensor container.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def kde(x: torch.Tensor, std: float = 0.1, half: bool = True, down=None) -> torch.Tensor:
    if not x.is_cuda:
        raise _lib.RomaHipError("kde: tensor must live on a HIP device; there is no CPU fallback")
    if x.dim() != 2 or x.shape[1] != 4:
        raise ValueError(f"kde: expected [n,4] matches (A-xy, B-xy), got {tuple(x.shape)}")
    lib = _lib.load()
    xs = x.detach().to(torch.float32).contiguous()
    n = xs.shape[0]
    density = torch.empty((n,), device=xs.device, dtype=torch.float32)
    if n == 0:
        return density
    stream = torch.cuda.current_stream(xs.device).cuda_stream
    with torch.cuda.device(xs.device):
        rc = lib.roma_op_kde(C.c_void_p(xs.data_ptr()), n, int(down) if down is not None else 1, float(std),
                             1 if half else 0, C.c_void_p(density.data_ptr()), C.c_void_p(stream))
    _lib.check(rc)
    return density
