"""Local-window correlation at the model's real shapes, in the two warp regimes that decide its traffic (GPU box).

  coherent   neighbouring queries look at neighbouring places (what a trained matcher produces): the windows of an
             8 x 8 query tile overlap, the tiled kernel stages their union once in LDS -> HBM traffic ~ algorithmic bytes
  incoherent every query looks somewhere else (the random-weight benchmark model: neighbouring coarse matches differ by
             ~12 of 40 tokens): each query needs its own (2r+2)^2 x C patch, (2r+2)^2 x the algorithmic bytes from
             L2 / Infinity Cache whatever the kernel does -> the tiles go to the per-query gather work list

Modes (roma_tuning "lc_mode"): 0 = tiled (16-bit features: all-pairs on the matrix core, round 3) + work list (default),
1 = every tile forced onto the gather work list (isolates the list kernel
against the per-pixel kernel on identical work), 2 = the per-pixel kernel of round 1.
Reports ms and ALGORITHMIC GB/s = (f0 + f1 read once + warp + outputs) / time.

    gpurun --timeout 300 -- 'python tools/bench_local_corr.py > gpurun_out/bench_local_corr.log 2>&1'
"""
import ctypes as C
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()
BF16, F32 = 1, 0


def P(t):
    return C.c_void_p(t.data_ptr())


def grid(B, h, w):
    ys = torch.linspace(-1 + 1 / h, 1 - 1 / h, h, device="cuda")
    xs = torch.linspace(-1 + 1 / w, 1 - 1 / w, w, device="cuda")
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack((gx, gy), dim=-1)[None].expand(B, h, w, 2).contiguous()


def run(r, c, h, w, B, regime, dt):
    tdt = torch.bfloat16 if dt == BF16 else torch.float32
    f = torch.randn(B, h, w, c, device="cuda").to(tdt)  # image b is the query image of pair b and the support of pair b + B/2
    K = (2 * r + 1) ** 2
    if regime == "coherent":
        warp = grid(B, h, w) * 0.93 + 0.03 + torch.randn(B, h, w, 2, device="cuda") * (0.3 / w)
    else:  # bilinear up-sampling of an incoherent coarse match, like the decoder's scale loop on random weights
        coarse = torch.rand(B, 2, 40, 40, device="cuda") * 2 - 1
        warp = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).contiguous()
    out = torch.empty(B, h * w, K, device="cuda", dtype=tdt)
    es = 2.0 if dt == BF16 else 4.0
    alg_bytes = B * h * w * (2.0 * c * es + 8.0 + K * es)
    res = {}
    outs = {}
    # (lc_mode, lc_bin): lc_bin = 1 sends the queries of incoherent tiles through the bin-sorted LIST form of the tile kernel
    # (round 6), 0 through per-query gathers (rounds 2-5)
    forms = {"tiled+binned (default)": (0, 1), "tiled+gather_list (r05 default)": (0, 0), "all_binned": (1, 1),
             "all_to_gather_list": (1, 0), "per_pixel": (2, 0)}
    for name, (mode, binned) in forms.items():
        lib.roma_tuning(b"lc_mode", mode)
        lib.roma_tuning(b"lc_bin", binned)

        def call():
            rc = lib.roma_op_local_corr_window(P(f), P(f), P(warp), P(out), B, h, w, c, r, c ** -0.5, K, dt, dt, None)
            assert rc == 0, lib.roma_last_error()
        out.fill_(float("nan"))
        call()
        torch.cuda.synchronize()
        outs[name] = out.float().clone()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        ms = statistics.median(ts)
        res[name] = {"ms": round(ms, 4), "algorithmic_GBs": round(alg_bytes / (ms * 1e-3) / 1e9)}
    ref = outs["per_pixel"]
    d = {k: float((v - ref).abs().max()) for k, v in outs.items() if k != "per_pixel"}
    lib.roma_tuning(b"lc_mode", -1)
    lib.roma_tuning(b"lc_bin", -1)
    print(json.dumps({"r": r, "C": c, "hw": [h, w], "B": B, "dtype": "bf16" if dt == BF16 else "f32", "warp": regime,
                      "algorithmic_MB": alg_bytes / 1e6, **res, "max_abs_diff_vs_per_pixel": d}), flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    regimes = tuple(sys.argv[1:]) or ("coherent", "incoherent")  # e.g. `bench_local_corr.py coherent` under rocprofv3 --pmc
    for regime in regimes:
        for (r, c, h, w) in ((7, 512, 40, 40), (3, 512, 70, 70), (3, 512, 108, 108), (2, 256, 140, 140), (2, 256, 216, 216)):
            run(r, c, h, w, 16, regime, BF16)
    run(3, 512, 108, 108, 16, "coherent", F32)
    run(2, 256, 216, 216, 16, "coherent", F32)
