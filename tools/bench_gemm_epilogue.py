"""The 16-bit GEMMs of one DINOv2 block and of the refiner's 1x1 convolutions with the epilogues the model uses (bias, bias +
GELU, bias + bf16 residual in place, q / k / V^T scatter): time per launch and a checksum of the output bits, so that two builds
(ROMA_LIB_DIR=... selects another library directory) can be compared for speed AND bit-identity on one box.

    python tools/bench_gemm_epilogue.py
"""
import ctypes as C
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()
BF16 = 1


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, iters=20):
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


def digest(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.cpu().view(torch.int16).numpy().tobytes())
    return h.hexdigest()[:12]


def case(tag, M, N, K, kind):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    outs = (out,)
    if kind in ("bias", "gelu", "relu"):
        act = {"bias": 0, "relu": 1, "gelu": 2}[kind]

        def fn():
            rc = lib.roma_op_gemm(P(A), K, P(W), K, P(out), N, M, N, K, 1, 0, 0, 0, P(b), None, None, 0, act, 1.0, BF16, BF16, None)
            assert rc == 0, _lib.last_error(lib)
    elif kind == "res":
        def fn():
            out.copy_(res)
            rc = lib.roma_op_gemm_res_bf16(P(A), K, P(W), K, P(out), N, M, N, K, P(b), None, P(out), N, None)
            assert rc == 0, _lib.last_error(lib)
    elif kind == "qkv":
        B, heads, hd = 16, 16, 64
        ntok = M // B
        npad = (ntok + 63) // 64 * 64
        q = torch.zeros(B, heads, npad, hd, device="cuda", dtype=torch.bfloat16)
        k = torch.zeros_like(q)
        vt = torch.zeros(B, heads, hd, npad, device="cuda", dtype=torch.bfloat16)
        outs = (q, k, vt)

        def fn():
            rc = lib.roma_op_qkv_scatter_gemm(P(A), P(W), P(b), P(q), P(k), P(vt), B, ntok, npad, heads, hd, K, BF16, BF16, None)
            assert rc == 0, _lib.last_error(lib)
    t = timeit(fn)
    if kind == "res":  # the copy is part of fn: time it alone and subtract
        t -= timeit(lambda: out.copy_(res))
    fn()
    torch.cuda.synchronize()
    print(f"{tag:34s} M={M:7d} N={N:5d} K={K:5d} {kind:5s} {t * 1e3:8.1f} us {2.0 * M * N * K / 1e9 / t:6.0f} TFLOP/s  sha1 {digest(*outs)}", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), "lib dir:", os.environ.get("ROMA_LIB_DIR", "(in tree)"))
    case("DINOv2 fc1 + GELU", 25616, 4096, 1024, "gelu")
    case("DINOv2 fc2 + residual", 25616, 1024, 4096, "res")
    case("DINOv2 proj + residual", 25616, 1024, 1024, "res")
    case("DINOv2 qkv scatter", 25616, 3072, 1024, "qkv")
    case("DINOv2 fc1 (ragged M) + GELU", 21920, 4096, 1024, "gelu")
    case("refiner 1x1 stride 4 pass 2", 746496, 576, 576, "relu")
    case("refiner 1x1 stride 8 pass 2", 186624, 1152, 1152, "relu")
    case("refiner 1x1 stride 16", 25600, 1408, 1408, "relu")
    case("decoder linear", 25600, 1024, 1536, "bias")
