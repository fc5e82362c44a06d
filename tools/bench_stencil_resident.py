"""Is the depthwise 5x5 stencil bound by HBM or by its own instruction stream?  (GPU box.)

The same kernels on (a) the benchmark's tensors (hundreds of MB: every byte comes from / goes to HBM) and (b) a tensor
small enough to live in the 4 MiB L2 of every XCD / the 256 MiB Infinity Cache (same per-pixel work, no HBM traffic after
the first launch).  If (b) is not much faster per element than (a), the kernel is paying for VALU issue / latency, not for
bytes - and moving its arithmetic into the producer of a GEMM's A operand cannot make it free: the same VALU work would
have to be issued by the GEMM's own waves, which run at the register / LDS limit.  Used for DESIGN.md section 4
("why the dw5x5 is not fused into the 1x1 GEMM at C >= 576").

    python tools/bench_stencil_resident.py
"""
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


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def case(tag, B, H, W, Cp, iters):
    x = torch.randn(B, H, W, Cp, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.randn(25, Cp, device="cuda") * 0.1
    b = torch.randn(Cp, device="cuda") * 0.1
    n = B * H * W * Cp

    def dw():
        assert lib.roma_op_dwconv5x5(P(x), P(y), P(w), P(b), B, H, W, Cp, BF16, None) == 0, lib.roma_last_error()
    t = timeit(dw, iters)
    print(f"dwconv5x5 {tag:34s} {B:2d}x{H:4d}x{W:4d}x{Cp:5d}  {n * 4 / 2**20:8.1f} MiB r+w  {t * 1e3:8.1f} us  "
          f"{n * 4 / t / 1e6:7.0f} GB/s  {n / t / 1e6:7.2f} Gelem/s  {n * 25 / t / 1e9:6.1f} TMAC/s", flush=True)
    if Cp in (24, 144):
        pw = (torch.randn(Cp, Cp, device="cuda") * 0.05).to(torch.bfloat16)
        pb = torch.randn(Cp, device="cuda")

        def rb():
            assert lib.roma_op_refiner_block(P(x), P(y), P(w), P(b), P(pw), P(pb), B, H, W, Cp, BF16, None) == 0, lib.roma_last_error()
        t = timeit(rb, iters)
        print(f"refiner_block {tag:30s} {B:2d}x{H:4d}x{W:4d}x{Cp:5d}  {n * 4 / 2**20:8.1f} MiB r+w  {t * 1e3:8.1f} us  "
              f"{n * 4 / t / 1e6:7.0f} GB/s  {n / t / 1e6:7.2f} Gelem/s", flush=True)


if __name__ == "__main__":
    # the benchmark's launches (HBM resident)
    case("bench stride 4 pass 2 (HBM)", 16, 216, 216, 576, 10)
    case("bench stride 8 pass 2 (HBM)", 16, 108, 108, 1152, 10)
    case("bench stride 2 pass 2 (HBM)", 16, 432, 432, 144, 10)
    # the same work per pixel on cache-resident tensors (2 x 10-40 MiB: Infinity Cache; 2 x 1-3 MiB per XCD: L2)
    case("Infinity-Cache resident", 2, 108, 108, 576, 200)
    case("Infinity-Cache resident", 2, 76, 76, 1152, 200)
    case("Infinity-Cache resident", 2, 216, 216, 144, 200)
    case("L2 resident", 1, 72, 72, 576, 500)
    case("L2 resident", 1, 52, 52, 1152, 500)
    case("L2 resident", 1, 144, 144, 144, 500)
