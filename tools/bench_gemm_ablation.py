"""What bounds the K loop of the 8-phase bf16 GEMM (gemm8p.hip)?  Compile-time ablation builds of the <bf16, dense, none>
kernel, both K-loop schedules (roma_tuning "gemm8p_sched": 0 = quadrant phases, reads 12 / 4 / 8 / 0; 1 = k-half phases,
reads 8 / 6 / 6 / 4), no epilogue (gemm_dbg bit 256) so that only the loop is timed:

    full            reads + LDS-DMA + MFMA (+ trace stamps)
    no_reads        LDS-DMA + MFMA                 -> what the fragment reads cost
    no_dma          reads + MFMA                   -> what the LDS-DMA (issue + LDS writes + L2 traffic) costs
    no_mfma         reads + LDS-DMA                -> the data-movement skeleton alone
    mfma_only       barriers + MFMA                -> the structure's ceiling
    dma_only / reads_only

and the phase trace of the full build: s_memtime at the first barrier release of each phase, waves 0 (group 0) and 4
(group 1) of 16 workgroups -> median cycles per phase P1..P4 and per K tile.

    gpurun --timeout 300 -- 'python tools/bench_gemm_ablation.py > gpurun_out/gemm_ablation.log 2>&1'
"""
import ctypes as C
import json
import os
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()
BF16 = 1
TRACE, NOEPI = 32768, 256
NO_RD, NO_DMA, NO_MF = 4096, 8192, 16384
VARIANTS = [("full", 0), ("no_reads", NO_RD), ("no_dma", NO_DMA), ("no_mfma", NO_MF), ("mfma_only", NO_RD | NO_DMA),
            ("dma_only", NO_RD | NO_MF), ("reads_only", NO_DMA | NO_MF)]


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


def trace_stats(nk, ntile_wg):
    buf = np.zeros(16 * 2 * 256 * 4, dtype=np.uint32)
    n = lib.roma_debug_gemm_trace(C.c_void_p(buf.ctypes.data), buf.nbytes)
    assert n == buf.nbytes, lib.roma_last_error()
    t = buf.reshape(16, 2, 256, 4).astype(np.int64)
    out = {}
    for grp in (0, 1):
        x = t[:, grp]                                     # [wg, ktile, phase]
        flat = x.reshape(16, -1)                          # stamps in time order: kt0 P1..P4, kt1 P1..P4, ...
        d = (flat[:, 1:] - flat[:, :-1]) & 0xFFFFFFFF     # cycles from one phase's release to the next
        nkt = min(256, nk * ntile_wg)                     # K tiles this workgroup really ran (and recorded)
        d = d[:, : 4 * nkt - 1]
        # steady state: skip the first 2 K tiles of every output tile and the tile seams (every nk K tiles)
        per_phase = [[], [], [], []]
        for i in range(d.shape[1]):
            kt, ph = divmod(i, 4)
            if kt % nk < 2 or kt % nk >= nk - 1:
                continue
            per_phase[ph].extend(d[:, i].tolist())
        med = [float(np.median(p)) if p else None for p in per_phase]
        out[f"group{grp}_cycles_P1..P4"] = med
        out[f"group{grp}_cycles_per_ktile"] = sum(m for m in med if m is not None)
    return out


def shape(M, N, K):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)

    def call():
        rc = lib.roma_op_gemm(P(A), K, P(W), K, P(out), N, M, N, K, 1, 0, 0, 0, None, None, None, 0, 0, 1.0, BF16, BF16, None)
        assert rc == 0, lib.roma_last_error()
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = -(-tiles // 256)
    res = {"shape": [M, N, K], "tiles": tiles, "rounds": rounds, "ideal_loop_us": round(2.0 * 256 * 256 * K / (2.5e15 / 256) * 1e6 * rounds, 2)}
    lib.roma_tuning(b"gemm8p", 1)
    for sched in (0, 1):
        lib.roma_tuning(b"gemm8p_sched", sched)
        r = {}
        # production builds first: whole kernel and loop only
        lib.roma_tuning(b"gemm_dbg", 0)
        r["prod_us"] = round(timed(call), 1)
        r["prod_TFLOPs"] = round(2.0 * M * N * K / (r["prod_us"] * 1e-6) / 1e12, 1)
        lib.roma_tuning(b"gemm_dbg", NOEPI)
        r["prod_noepi_us"] = round(timed(call), 1)
        for name, bits in VARIANTS:
            lib.roma_tuning(b"gemm_dbg", TRACE | NOEPI | bits)
            r[f"{name}_us"] = round(timed(call), 1)
            if name == "full":
                call()
                torch.cuda.synchronize()
                r["trace"] = trace_stats(K // 64, tiles // 256 if tiles >= 256 else 1)
        res[f"sched{sched}"] = r
    lib.roma_tuning(b"gemm_dbg", -1)
    lib.roma_tuning(b"gemm8p_sched", -1)
    lib.roma_tuning(b"gemm8p", -1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    shape(8192, 8192, 8192)      # the canonical yardstick: 1024 tiles = 4 rounds, 128 K tiles each
    shape(16384, 1024, 4096)     # one round, long K (fc2-like)
    shape(16384, 1024, 1024)     # one round, K = 1024 (proj-like)
    shape(25616, 4096, 1024)     # fc1 shape
    shape(25616, 1024, 4096)     # fc2 shape
