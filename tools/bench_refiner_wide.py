"""The C = 576 ConvRefiner block at the benchmark's shapes: the fused kernel (refiner_block_wide.hip) against the pair it
replaces (dwconv5x5 ring kernel + weight-stationary 1x1 GEMM), same operands, bf16 equality statistics of the two results.
    python tools/bench_refiner_wide.py [--reps 20]"""
import argparse
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


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(B, H, W, Cp, reps):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, H, W, Cp, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(25, Cp, device="cuda", generator=g) * 0.1
    b = torch.randn(Cp, device="cuda", generator=g) * 0.1
    pw = (torch.randn(Cp, Cp, device="cuda", generator=g) * Cp ** -0.5).to(torch.bfloat16)
    pb = torch.randn(Cp, device="cuda", generator=g)
    y1, t, y2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    M = B * H * W

    def fused():
        assert lib.roma_op_refiner_block(P(x), P(y1), P(w), P(b), P(pw), P(pb), B, H, W, Cp, BF16, None) == 0, _lib.last_error(lib)

    def dw():
        assert lib.roma_op_dwconv5x5(P(x), P(t), P(w), P(b), B, H, W, Cp, BF16, None) == 0

    def pwc():
        assert lib.roma_op_gemm(P(t), Cp, P(pw), Cp, P(y2), Cp, M, Cp, Cp, 1, 0, 0, 0, P(pb), None, None, 0, 0, 1.0, BF16, BF16, None) == 0

    us_f = timed(fused, reps)
    lib.roma_tuning(b"rb_wide", 2)  # the stencil as pairs of v_fma_f32 instead of v_pk_fma_f32
    us_f2 = timed(fused, reps)
    lib.roma_tuning(b"rb_wide", -1)
    us_d, us_p = timed(dw, reps), timed(pwc, reps)
    diff = (y1.float() - y2.float()).abs()
    same = float((y1.view(torch.int16) == y2.view(torch.int16)).float().mean())
    fl = 2.0 * M * Cp * Cp
    print(f"B{B} {H}x{W} C={Cp}: fused {us_f:8.1f} us [scalar-FMA stencil {us_f2:8.1f}] ({fl / us_f * 1e-6:6.0f} TFLOP/s of the 1x1, {4.0 * M * Cp / us_f * 1e-6:5.2f} TB/s in+out) | "
          f"dwconv {us_d:7.1f} + 1x1 {us_p:7.1f} = {us_d + us_p:8.1f} us | x{(us_d + us_p) / us_f:5.2f} | "
          f"identical bf16 {100 * same:6.2f} %, max |diff| {float(diff.max()):.3g} (|y| max {float(y2.float().abs().max()):.3g})", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    for (B, H, W) in ((16, 216, 216), (8, 216, 216), (16, 140, 140), (8, 140, 140), (1, 140, 140)):
        run(B, H, W, 576, a.reps)
