"""conv64.hip against the implicit GEMM on the three VGG layers it takes (conv1_2: 64 -> 64, conv2_1: 64 -> 128, conv2_2: 128 -> 128) at the
bench's sizes (16 images: 560^2 / 864^2 and the pooled halves).  HIP-event timing, inputs resident.
    python tools/bench_conv64.py
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from roma_amd import _lib  # noqa: E402


def P(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def main():
    lib = _lib.load()
    torch.manual_seed(0)
    for (B, H, W, Cin, Cout) in [(16, 560, 560, 64, 64), (16, 280, 280, 64, 128), (16, 280, 280, 128, 128), (16, 864, 864, 64, 64),
                                 (16, 432, 432, 64, 128), (16, 432, 432, 128, 128)]:
        x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
        w = (torch.randn(Cout, 9 * Cin, device="cuda") / (3 * Cin ** 0.5)).bfloat16()
        b = torch.randn(Cout, device="cuda")
        out = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)
        flop = 2.0 * B * H * W * Cout * 9 * Cin
        for mode in (0, 3):
            lib.roma_tuning(b"conv64", mode)
            for _ in range(3):
                lib.roma_op_conv3x3(P(x), P(w), P(b), P(out), B, H, W, Cin, Cout, 1, 1, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                lib.roma_op_conv3x3(P(x), P(w), P(b), P(out), B, H, W, Cin, Cout, 1, 1, None)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            print(f"B{B} {H}x{W} {Cin}->{Cout} conv64={mode}: {us:9.1f} us  {flop / us / 1e6:7.1f} TF/s", flush=True)
        lib.roma_tuning(b"conv64", -1)


def first_layer(lib):
    torch.manual_seed(0)
    for (B, H, W) in [(8, 560, 560), (8, 864, 864)]:
        img = torch.randn(B, 3, H, W, device="cuda")
        w = torch.zeros(64, 32, device="cuda", dtype=torch.bfloat16)
        w[:, :27] = (torch.randn(64, 27, device="cuda") / 5).bfloat16()
        b = torch.randn(64, device="cuda")
        out = torch.empty(B, H, W, 64, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            lib.roma_op_conv3x3_c3_bf16(P(img), P(w), P(b), P(out), B, H, W, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            lib.roma_op_conv3x3_c3_bf16(P(img), P(w), P(b), P(out), B, H, W, None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"first layer B{B} {H}x{W}: {us:9.1f} us  {B * H * W * 140 / us / 1e6:6.2f} TB/s algorithmic", flush=True)


if __name__ == "__main__":
    first_layer(_lib.load())
    main()
