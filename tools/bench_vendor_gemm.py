"""Yardstick only (never on the product path): what does the vendor GEMM (torch.nn.functional.linear on ROCm = hipBLASLt /
rocBLAS) reach on the model's GEMM shapes, next to libroma_hip's kernels on the same operands?  Answers "how far are the
hand-written loops from a tuned library on THESE shapes" for DESIGN.md; the library is not linked, not loaded and not
called by roma_amd.

    python tools/bench_vendor_gemm.py
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


def timeit(fn, iters=10):
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


def case(tag, M, N, K):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    bb = b.to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def ours():
        rc = lib.roma_op_gemm(P(A), K, P(W), K, P(out), N, M, N, K, 1, 0, 0, 0, P(b), None, None, 0, 0, 1.0, BF16, BF16, None)
        assert rc == 0, lib.roma_last_error()

    def vendor():
        torch.nn.functional.linear(A, W, bb)

    def vendor_nobias():
        torch.matmul(A, W.t())
    fl = 2.0 * M * N * K / 1e9
    to, tv, tn = timeit(ours), timeit(vendor), timeit(vendor_nobias)
    print(f"{tag:34s} M={M:7d} N={N:5d} K={K:5d}  libroma_hip {to*1e3:7.1f} us {fl/to:6.0f} TFLOP/s | vendor linear+bias {tv*1e3:7.1f} us "
          f"{fl/tv:6.0f} | vendor matmul {tn*1e3:7.1f} us {fl/tn:6.0f}", flush=True)


if __name__ == "__main__":
    print(torch.__version__, torch.cuda.get_device_name(0))
    case("refiner 1x1 stride 4 pass 2", 746496, 576, 576)
    case("refiner 1x1 stride 4 pass 1", 313600, 576, 576)
    case("refiner 1x1 stride 8 pass 2", 186624, 1152, 1152)
    case("refiner 1x1 stride 8 pass 1", 78400, 1152, 1152)
    case("refiner 1x1 stride 16", 25600, 1408, 1408)
    case("DINOv2 fc1", 25616, 4096, 1024)
    case("DINOv2 fc2", 25616, 1024, 4096)
    case("DINOv2 proj", 25616, 1024, 1024)
    case("DINOv2 qkv (plain)", 25616, 3072, 1024)
    case("single pair fc1", 3202, 4096, 1024)
    case("single pair fc2", 3202, 1024, 4096)
    case("square 8192", 8192, 8192, 8192)
