"""Where does a GEMM tile's time go?  Both bf16 main loops (gemm8p / classic) on model shapes with the tuning bits of
roma_tuning("gemm_dbg"): 1 = no global stores in the epilogue, 2 = minimal K loop (classic: 1 slab, gemm8p: 2 K tiles),
2048 = NO non-temporal output stores (they are the default: bf16 row writer, full tiles), 256 = no epilogue at all (gemm8p / gemm6p), 128 = no wave-group stagger (gemm8p / gemm6p), 512 = LDS-DMA issued between
the MFMAs instead of in the load block (gemm8p, plain bf16 epilogue only).  Widths that tile better by 192 than by 256 run
on gemm6p.hip when "gemm8p" is on.

    t(full) - t(bits 2)        ~ K-loop time        t(bits 2) ~ per-tile overhead (prologue + epilogue + launch)
    t(full) - t(bits 1)        ~ cost of the global stores
    t(full) - t(bits 256)      ~ whole epilogue (gemm8p)

    gpurun --timeout 300 -- 'python tools/bench_gemm_overhead.py > gpurun_out/bench_gemm_overhead.log 2>&1'
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
BF16 = 1


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def timed(fn, iters=5, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)  # us
    return statistics.median(ts)


def shape(M, N, K, act=0):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)

    def call():
        rc = lib.roma_op_gemm(P(A), K, P(W), K, P(out), N, M, N, K, 1, 0, 0, 0, P(b), None, None, 0, act, 1.0, BF16, BF16, None)
        assert rc == 0, lib.roma_last_error()
    res = {"shape": [M, N, K], "act": act, "tiles_256": ((M + 255) // 256) * ((N + 255) // 256)}
    for kern, mode in (("classic", 0), ("gemm8p", 1)):
        lib.roma_tuning(b"gemm8p", mode)
        r = {}
        for bits in (0, 1, 2, 3) + ((128, 256, 258, 512, 2048) if mode else ()):
            lib.roma_tuning(b"gemm_dbg", bits)
            r[f"dbg{bits}_us"] = round(timed(call), 1)
        lib.roma_tuning(b"gemm_dbg", 0)
        r["TFLOPs"] = round(2.0 * M * N * K / (r["dbg0_us"] * 1e-6) / 1e12, 1)
        res[kern] = r
    lib.roma_tuning(b"gemm8p", -1)
    lib.roma_tuning(b"gemm_dbg", -1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    shape(25600, 1024, 1024)          # 400 tiles: 2 rounds on 256 CUs
    shape(16384, 1024, 1024)          # 256 tiles: exactly one round
    shape(32768, 1024, 1024)          # 512 tiles: exactly two rounds
    shape(25616, 4096, 1024, act=2)   # fc1 + GELU
    shape(25616, 1024, 4096)          # fc2
    shape(16384, 1024, 4096)          # one round, long K
    shape(78400, 1152, 1152)          # stride-8 refiner 1x1 (256 x 192 tiles: classic vs gemm6p)
    shape(313600, 576, 576)           # stride-4 refiner 1x1 (256 x 192 tiles: classic vs gemm6p)
