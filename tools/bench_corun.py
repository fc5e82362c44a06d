"""What do two kernel classes gain from sharing the chip?  Two HIP streams, each running one operator in a loop; per pair
(A on stream 1, B on stream 2): time of A alone, of B alone, and of both at once for the same number of launches.  A persistent
GEMM owns the CUs it runs on, so the two streams interleave at workgroup granularity: efficiency = (tA + tB) / t(both) says whether
pairing DIFFERENT classes (matrix-core GEMM beside a VALU / HBM-bound stencil) beats running each on the full chip in turn -
tools/bench_gemm_burst.py shows a GEMM K tile taking 1.82 us with 256 workgroups on the chip and 1.32 us with 128.

    python tools/bench_corun.py
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


def gemm(M, N, K, act):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)

    def fn(s):
        rc = lib.roma_op_gemm(P(A), K, P(W), K, P(out), N, M, N, K, 1, 0, 0, 0, P(b), None, None, 0, act, 1.0, BF16, BF16, C.c_void_p(s))
        assert rc == 0, _lib.last_error(lib)
    return fn


def dwconv(B, H, Cp):
    x = torch.randn(B, H, H, Cp, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.randn(25, Cp, device="cuda") * 0.1
    b = torch.randn(Cp, device="cuda") * 0.1

    def fn(s):
        assert lib.roma_op_dwconv5x5(P(x), P(y), P(w), P(b), B, H, H, Cp, 1, C.c_void_p(s)) == 0
    return fn


def block(B, H, Cp):
    x = torch.randn(B, H, H, Cp, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.randn(25, Cp, device="cuda") * 0.1
    b = torch.randn(Cp, device="cuda") * 0.1
    pw = (torch.randn(Cp, Cp, device="cuda") * 0.05).to(torch.bfloat16)
    pb = torch.randn(Cp, device="cuda")

    def fn(s):
        assert lib.roma_op_refiner_block(P(x), P(y), P(w), P(b), P(pw), P(pb), B, H, H, Cp, 1, C.c_void_p(s)) == 0
    return fn


def attention(B, heads, hd, N):
    npad = (N + 127) // 128 * 128
    q = (torch.randn(B, heads, npad, hd, device="cuda") * 0.15).to(torch.bfloat16)
    k = torch.randn(B, heads, npad, hd, device="cuda").to(torch.bfloat16)
    vt = torch.randn(B, heads, hd, npad, device="cuda").to(torch.bfloat16)
    o = torch.zeros(B * N, heads * hd, device="cuda", dtype=torch.bfloat16)

    def fn(s):
        assert lib.roma_op_attention(P(q), P(k), P(vt), P(o), B, heads, N, npad, hd, 1, 1, C.c_void_p(s)) == 0
    return fn


def wall(jobs):
    """jobs: [(fn, stream, n)] - all launched, then one device sync; ms"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    e0.record(main)
    for _, s, _ in jobs:
        s.wait_stream(main)
    nmax = max(n for _, _, n in jobs)
    for i in range(nmax):  # interleaved submission
        for fn, s, n in jobs:
            if i < n:
                fn(s.cuda_stream)
    for _, s, _ in jobs:
        main.wait_stream(s)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ops = {
        "fc1 GEMM (half batch)": gemm(12808, 4096, 1024, 2),
        "fc2 GEMM (half batch)": gemm(12808, 1024, 4096, 0),
        "dwconv 216^2 x 576 (B=8)": dwconv(8, 216, 576),
        "block C=144 432^2 (B=8)": block(8, 432, 144),
        "attention 8 x 16 x 1601": attention(8, 16, 64, 1601),
        "1x1 C=1152 (B=8)": gemm(93312, 1152, 1152, 1),
    }
    names = list(ops)
    for f in ops.values():
        for _ in range(3):
            f(s1.cuda_stream)
    torch.cuda.synchronize()
    alone = {}
    for n in names:
        t = wall([(ops[n], s1, 20)])
        alone[n] = t / 20
        print(f"alone  {n:28s} {alone[n] * 1e3:8.1f} us")
    pairs = [(0, 0), (0, 2), (0, 3), (0, 4), (1, 2), (5, 2), (5, 3), (2, 2), (3, 3), (2, 3), (4, 2)]
    for ia, ib in pairs:
        a, b = names[ia], names[ib]
        # equal time on both streams: launches inversely proportional to the time alone
        T = 8.0  # ms of work per stream
        na, nb = max(1, round(T / alone[a])), max(1, round(T / alone[b]))
        ta, tb = wall([(ops[a], s1, na)]), wall([(ops[b], s2, nb)])
        both = min(wall([(ops[a], s1, na), (ops[b], s2, nb)]) for _ in range(2))
        print(f"{a:28s} x{na:3d} + {b:28s} x{nb:3d}: alone {ta:6.2f} + {tb:6.2f} = {ta + tb:6.2f} ms, together {both:6.2f} ms, efficiency {(ta + tb) / both:5.3f}", flush=True)
