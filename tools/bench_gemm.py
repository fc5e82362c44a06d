"""Micro-benchmark of roma_op_gemm / conv3x3 shapes (GPU box): separates main-loop from epilogue cost."""
import ctypes as C
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()
F32, BF16 = 0, 1


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def run(M, N, K, dt_in, dt_out, iters=10, bias=True):
    tin = torch.float32 if dt_in == F32 else torch.bfloat16
    tout = torch.float32 if dt_out == F32 else torch.bfloat16
    A = torch.randn(M, K, device="cuda").to(tin)
    W = torch.randn(N, K, device="cuda").to(tin)
    b = torch.randn(N, device="cuda") if bias else None
    out = torch.empty(M, N, device="cuda", dtype=tout)
    def call():
        rc = lib.roma_op_gemm(P(A), K, P(W), K, P(out), N, M, N, K, 1, 0, 0, 0, P(b), None, None, 0, 0, 1.0, dt_in, dt_out, None)
        assert rc == 0, lib.roma_last_error()
    call(); call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    print(f"M={M:7d} N={N:5d} K={K:5d} in={'f32' if dt_in==F32 else 'bf16'} out={'f32' if dt_out==F32 else 'bf16'}: {ms:8.3f} ms  {tf:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    for (M, N, K) in [(65536, 1152, 1152), (65536, 1152, 9216), (65536, 1024, 1024), (65536, 1024, 8192), (65536, 576, 576),
                      (65536, 128, 1152), (25616, 3072, 1024), (25616, 4096, 1024), (25616, 1024, 4096), (65536, 144, 144), (262144, 24, 24)]:
        run(M, N, K, BF16, BF16)
        run(M, N, K, BF16, F32)
    for (M, N, K) in [(65536, 1152, 1152), (65536, 1152, 4608), (25616, 1024, 4096)]:
        run(M, N, K, F32, F32)
