"""The VGG layers with Cout >= 256 at the benchmark's shapes: patch-resident kernel (conv_patch.hip) against the implicit GEMM
(gemm8p conv form) on the same slab-major weights, interleaved in one process."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()


def P(t):
    return C.c_void_p(t.data_ptr())


def run(B, H, W, Cin, Cout):
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device="cuda")
    out = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    flops = 2.0 * B * H * W * Cout * 9 * Cin
    res = {}
    keep = {}
    for rnd_ in range(3):
        for patch in (1, 0):
            lib.roma_tuning(b"conv_patch", patch)
            for _ in range(2):
                assert lib.roma_op_conv3x3_slab(P(x), P(w), P(b), P(out), B, H, W, Cin, Cout, 1, 1, None) == 0, lib.roma_last_error()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                lib.roma_op_conv3x3_slab(P(x), P(w), P(b), P(out), B, H, W, Cin, Cout, 1, 1, None)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(patch, []).append(e0.elapsed_time(e1) * 1e3 / n)
            keep[patch] = out.clone()
    lib.roma_tuning(b"conv_patch", -1)
    same = bool(torch.equal(keep[0].view(torch.int16), keep[1].view(torch.int16)))
    t1, t0 = min(res[1]), min(res[0])
    print(f"B{B} {H}x{W} {Cin}->{Cout}: patch {t1:8.1f} us {flops / t1 / 1e6:7.0f} TFLOP/s | implicit GEMM {t0:8.1f} us {flops / t0 / 1e6:7.0f} TFLOP/s"
          f" | x{t0 / t1:5.2f}  bit-identical {same}", flush=True)


if __name__ == "__main__":
    for (h, w) in ((216, 216), (140, 140)):
        run(16, h, w, 128, 256)
        run(16, h, w, 256, 256)
    for (h, w) in ((108, 108), (70, 70)):
        run(16, h, w, 256, 512)
        run(16, h, w, 512, 512)
    run(8, 216, 216, 256, 256)
    run(8, 108, 108, 512, 512)
