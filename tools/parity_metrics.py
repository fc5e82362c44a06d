"""Parity metrics shared by tests/ and bench.py (numpy only; no oracle, no reference import).

The outputs of match() (romatch/models/matcher.py:904-929) are `warp [B,H,2W,4]` and `certainty [B,H,2W]` in the
symmetric layout: the LEFT half holds A->B (grid in channels 0:2, B coordinates in 2:4), the RIGHT half holds B->A
(A coordinates in 0:2, grid in 2:4).  The grid channels are constants, identical in every implementation, so every
error statistic here is taken over the FLOW channels only (a median over all four channels is dominated by exact
zeros and says nothing).

The coarse match is an arg-max over 4096 class logits (utils/utils.py:300-322): it is discontinuous, so a
reduced-precision run may legitimately pick the runner-up class where the reference's top-2 logit gap is tiny.  Such
tokens are counted (`coarse_flips`) and must all lie below a stated gap; the continuous rest of the pipeline is held to
a bound with the reference's coarse match injected (roma_debug_inject).
"""
from __future__ import annotations

import numpy as np


def flow_channels(warp: np.ndarray, symmetric: bool = True) -> np.ndarray:
    """[B,H,Wt,4] -> [B,H,Wt,2]: the predicted coordinates (not the constant grid)."""
    if not symmetric:
        return warp[..., 2:]
    w = warp.shape[2] // 2
    return np.concatenate([warp[:, :, :w, 2:], warp[:, :, w:, :2]], axis=2)


def _stats(e: np.ndarray, tol: float) -> dict:
    e = np.asarray(e, dtype=np.float64).ravel()
    return {"max": float(e.max()), "p99": float(np.percentile(e, 99)), "p50": float(np.percentile(e, 50)),
            "mean": float(e.mean()), "frac_over_tol": float((e > tol).mean())}


def output_errors(warp, cert, ref_warp, ref_cert, symmetric: bool = True, tol: float = 1e-3) -> dict:
    """Error statistics of (warp, certainty) against a reference on the same (sub-sampled) grid."""
    ew = np.abs(flow_channels(np.asarray(warp), symmetric) - flow_channels(np.asarray(ref_warp), symmetric))
    ec = np.abs(np.asarray(cert) - np.asarray(ref_cert))
    grid_err = float(np.abs(np.asarray(warp) - np.asarray(ref_warp)).max() - ew.max()) if ew.size else 0.0
    return {"flow": _stats(ew, tol), "cert": _stats(ec, tol), "grid_channels_exact": bool(grid_err <= 0.0)}


def coarse_flips(gm_flow16: np.ndarray, ref_gm_flow16: np.ndarray, ref_gap: np.ndarray, thresh: float = 0.02) -> dict:
    """Tokens whose coarse match left the reference's class.

    gm_flow16 / ref: [b, T, 2] (any matching layout), ref_gap: the reference's top-1 minus top-2 class logit per token.
    Inside one class the refined coarse flow moves by far less than half a cell (1/64 of the [-1,1] range); a change
    of arg-max class moves it by at least ~1/32, so |d| > thresh separates the two cases."""
    d = np.abs(np.asarray(gm_flow16, np.float64) - np.asarray(ref_gm_flow16, np.float64)).reshape(-1, 2).max(axis=1)
    gap = np.asarray(ref_gap, np.float64).ravel()
    flipped = d > thresh
    return {"tokens": int(d.size), "flips": int(flipped.sum()),
            "max_gap_of_flipped": float(gap[flipped].max()) if flipped.any() else 0.0,
            "max_flow16_err_unflipped": float(d[~flipped].max()) if (~flipped).any() else 0.0}


def nchw_to_tokens(x: np.ndarray) -> np.ndarray:
    """[b,c,h,w] -> [b,h*w,c] (the C ABI's channels-last token layout)."""
    b, c, h, w = x.shape
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1)).reshape(b, h * w, c))
