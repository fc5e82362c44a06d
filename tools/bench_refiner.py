"""Micro-benchmark of one ConvRefiner block (dw5x5+BN+ReLU, then 1x1) per scale of the 560->864 workload (GPU box).
Prints per-kernel ms and effective HBM GB/s so the per-scale cost of the refiners is visible."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()
BF16 = 1


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(B, H, W, Cp):
    x = torch.randn(B, H, W, Cp, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.randn(25, Cp, device="cuda") * 0.1
    b = torch.randn(Cp, device="cuda") * 0.1
    pw = (torch.randn(Cp, Cp, device="cuda") * 0.05).to(torch.bfloat16)
    pb = torch.randn(Cp, device="cuda")
    M = B * H * W

    def dw():
        rc = lib.roma_op_dwconv5x5(P(x), P(y), P(w), P(b), B, H, W, Cp, BF16, None)
        assert rc == 0, lib.roma_last_error()

    def gm():
        rc = lib.roma_op_gemm(P(y), Cp, P(pw), Cp, P(x), Cp, M, Cp, Cp, 1, 0, 0, 0, P(pb), None, None, 0, 0, 1.0, BF16, BF16, None)
        assert rc == 0, lib.roma_last_error()

    tdw, tgm = timeit(dw), timeit(gm)
    gb = 2.0 * M * Cp * 2 / 1e9
    tf = 2.0 * M * Cp * Cp / 1e12
    print(f"B={B:2d} {H:4d}x{W:4d} C={Cp:5d}: dw {tdw:7.3f} ms {gb/tdw*1e3:7.0f} GB/s | pw {tgm:7.3f} ms {gb/tgm*1e3:7.0f} GB/s "
          f"{tf/tgm*1e3:6.0f} TF/s | block x9 = {9*(tdw+tgm):7.2f} ms", flush=True)
    if hasattr(lib, "roma_op_refiner_block"):
        def fz():
            rc = lib.roma_op_refiner_block(P(x), P(y), P(w), P(b), P(pw), P(pb), B, H, W, Cp, BF16, None)
            assert rc == 0, lib.roma_last_error()
        try:
            tf_ = timeit(fz)
            print(f"{'':28s} fused {tf_:7.3f} ms {gb/tf_*1e3:7.0f} GB/s | x9 = {9*tf_:7.2f} ms", flush=True)
        except AssertionError as e:
            print("   fused: n/a", e)


if __name__ == "__main__":
    B = 16
    for (res, scales) in [(560, [(16, 1408), (8, 1152), (4, 576), (2, 144), (1, 24)]), (864, [(8, 1152), (4, 576), (2, 144), (1, 24)])]:
        for s, Cp in scales:
            run(B, res // s, res // s, Cp)
