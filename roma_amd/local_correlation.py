"""The reference's operator boundary for the fused local-correlation kernel, on HIP.

Mirrors
  * `local_corr.local_corr(feature0, feature1, warp, mode=..., normalized_coords=True)`
    - the external fused-local-corr wheel as called at romatch/utils/local_correlation.py:26-32
  * `local_correlation(feature0, feature1, local_radius, warp, use_custom_corr=...)`
    - romatch/utils/local_correlation.py:77-143
torch is only the tensor container; the arithmetic is `roma_op_local_corr[_window]` in libroma_hip.
"""
from __future__ import annotations

import ctypes as C
from typing import Literal

import torch

from . import _lib


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.ROMA_F32
    if t.dtype == torch.bfloat16:
        return _lib.ROMA_BF16
    if t.dtype == torch.float16:
        return _lib.ROMA_F16
    raise TypeError(f"unsupported dtype {t.dtype} (float32, bfloat16 or float16)")


def _lib_for(t: torch.Tensor):
    """float16 tensors go to the library built for IEEE binary16 storage, everything else to the bf16 one"""
    return _lib.load(_lib.fmt_of(t.dtype))


def _require_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.RomaHipError("local_corr: tensors must live on a HIP device; there is no CPU fallback")


def local_corr(feature0: torch.Tensor, feature1: torch.Tensor, warp: torch.Tensor,
               mode: Literal["bilinear", "nearest"] = "bilinear", normalized_coords: bool = True) -> torch.Tensor:
    """feature0 [B,HW,C] (pre-scaled), feature1 [B,H,W,C] channels-last, warp [B,HW,K,2] -> [B,HW,K] f32."""
    if mode not in ("bilinear", "nearest"):
        raise ValueError(f"mode must be 'bilinear' or 'nearest' (got {mode!r})")
    if not normalized_coords:
        raise NotImplementedError("only normalized_coords=True is implemented (local_correlation.py:31)")
    _require_cuda(feature0, feature1, warp)
    B, H, W, Cc = feature1.shape
    K = warp.shape[2]
    assert feature0.shape == (B, H * W, Cc) and warp.shape == (B, H * W, K, 2)
    f0, f1 = feature0.contiguous(), feature1.contiguous()
    wp = warp.float().contiguous()
    out = torch.empty((B, H * W, K), device=f0.device, dtype=torch.float32)
    stream = torch.cuda.current_stream(f0.device).cuda_stream
    lib = _lib_for(f0)
    with torch.cuda.device(f0.device):  # the C ABI launches on the CURRENT device: make it the tensors' device
        _lib.check(lib.roma_op_local_corr(C.c_void_p(f0.data_ptr()), C.c_void_p(f1.data_ptr()),
                                          C.c_void_p(wp.data_ptr()), C.c_void_p(out.data_ptr()), B, H, W, Cc, K,
                                          int(mode == "nearest"), _dt(f0), _lib.ROMA_F32, C.c_void_p(stream)), lib=lib)
    return out


def local_correlation(feature0: torch.Tensor, feature1: torch.Tensor, local_radius: int, warp: torch.Tensor, *,
                      use_custom_corr: bool = True, padding_mode="zeros",
                      sample_mode: Literal["bilinear", "nearest"] = "bilinear") -> torch.Tensor:
    """feature0/feature1 [B,C,H,W], warp [B,2,H,W] -> corr [B,(2r+1)^2,H,W]  (local_correlation.py:77-143).

    The window taps are exactly one f1 pixel apart, so the HIP kernel takes only the centre warp and
    evaluates all taps from one (2r+2)^2 integer patch (see csrc/local_corr.hip)."""
    assert padding_mode == "zeros"
    if sample_mode not in ("bilinear", "nearest"):
        raise ValueError(f"sample_mode must be 'bilinear' or 'nearest' (got {sample_mode!r})")
    _require_cuda(feature0, feature1, warp)
    B, c, h, w = feature0.shape
    r = int(local_radius)
    K = (2 * r + 1) ** 2
    if sample_mode == "nearest":
        # nearbyint of every tap's own coordinate: no shared fractional offset, so the taps are built exactly as the
        # reference's wrapper builds them (local_correlation.py:93-108, 24-32) and go through the per-tap operator
        lw = torch.meshgrid(torch.linspace(-2 * r / h, 2 * r / h, 2 * r + 1, device=warp.device),
                            torch.linspace(-2 * r / w, 2 * r / w, 2 * r + 1, device=warp.device), indexing="ij")
        lw = torch.stack((lw[1], lw[0]), dim=-1).reshape(1, 1, 1, K, 2)
        coords = (warp.permute(0, 2, 3, 1)[:, :, :, None].float() + lw).reshape(B, h * w, K, 2)
        f0s = (feature0.reshape(B, c, h * w).permute(0, 2, 1) / (c ** 0.5)).contiguous()
        out = local_corr(f0s, feature1.permute(0, 2, 3, 1).contiguous(), coords, mode="nearest")
        return out.permute(0, 2, 1).reshape(B, K, h, w)
    f0 = feature0.permute(0, 2, 3, 1).contiguous()
    f1 = feature1.permute(0, 2, 3, 1).contiguous()
    wp = warp.permute(0, 2, 3, 1).float().contiguous()
    out = torch.empty((B, h * w, K), device=f0.device, dtype=torch.float32)
    stream = torch.cuda.current_stream(f0.device).cuda_stream
    lib = _lib_for(f0)
    with torch.cuda.device(f0.device):
        _lib.check(lib.roma_op_local_corr_window(C.c_void_p(f0.data_ptr()), C.c_void_p(f1.data_ptr()),
                                                 C.c_void_p(wp.data_ptr()), C.c_void_p(out.data_ptr()), B, h, w, c, r,
                                                 1.0 / (c ** 0.5), K, _dt(f0), _lib.ROMA_F32, C.c_void_p(stream)), lib=lib)
    return out.permute(0, 2, 1).reshape(B, K, h, w)
