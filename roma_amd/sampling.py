"""torch.multinomial(weights, k, replacement=False) on the device (`roma_op_multinomial`): the two draws of
RegressionMatcher.sample / TinyRoMa.sample (romatch/models/matcher.py:615-627, tiny.py:259-273).

Exponential race + radix select in HIP (csrc/sampling.hip), no sort and no host synchronisation.  The seed of every draw
comes from torch's CPU generator, so `torch.manual_seed(s)` makes a sampling sequence reproducible; the stream of random
numbers is not torch's, parity with the reference is distributional (as for any multinomial on another device)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def multinomial(weights: torch.Tensor, num_samples: int, generator=None) -> torch.Tensor:
    """Indices [num_samples] (int64, distinct, in draw order like torch.multinomial) drawn without replacement with probability proportional to
    `weights` [n] (>= 0).  Like torch on a GPU, the number of positive weights is not checked (no host synchronisation):
    if fewer than num_samples are positive, zero-weight entries complete the sample."""
    if not weights.is_cuda:
        raise _lib.RomaHipError("roma_amd.multinomial: weights must live on a HIP device; there is no CPU fallback")
    if weights.dim() != 1:
        raise ValueError("roma_amd.multinomial: expected a 1-D weight vector")
    n, k = int(weights.shape[0]), int(num_samples)
    if k <= 0 or k > n:
        raise RuntimeError("cannot sample n_sample > prob_dist.size(-1) samples without replacement")
    lib = _lib.load()
    w = weights.detach().to(torch.float32).contiguous()
    seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())  # CPU generator: no device synchronisation
    dev = w.device
    out = torch.empty((k,), device=dev, dtype=torch.int64)
    nws = int(lib.roma_op_multinomial_workspace(n, k))
    ws = torch.empty((nws,), device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _lib.check(lib.roma_op_multinomial(C.c_void_p(w.data_ptr()), n, k, C.c_ulonglong(seed), C.c_void_p(out.data_ptr()),
                                           C.c_void_p(ws.data_ptr()), nws, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out
