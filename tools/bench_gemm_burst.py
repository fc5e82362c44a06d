"""What does a 256 x 256 tile's epilogue cost when FEWER workgroups store at the same time?

The persistent gemm8p grid is one workgroup per CU, all of them in lock-step: every round of tiles ends with 256 workgroups
writing 128 KB each (33.5 MB) at the same moment.  roma_tuning("gemm8p_maxwg", n) caps the grid, roma_tuning("gemm_dbg", 256)
removes the epilogue: per cap,

    epilogue cost per round and workgroup = (t(full) - t(no epilogue)) / rounds(cap)

If that cost falls with the number of concurrent writers, the epilogue is bound by the store burst (HBM / fabric), and
de-phasing the workgroups would hide it; if it stays, it is the workgroup's own issue work.

    python tools/bench_gemm_burst.py
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


def shape(tag, M, N, K, act):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)

    def call():
        rc = lib.roma_op_gemm(P(A), K, P(W), K, P(out), N, M, N, K, 1, 0, 0, 0, P(b), None, None, 0, act, 1.0, BF16, BF16, None)
        assert rc == 0, _lib.last_error(lib)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    for cap in (256, 128, 64, 32):
        lib.roma_tuning(b"gemm8p_maxwg", cap)
        rounds = -(-tiles // cap)
        lib.roma_tuning(b"gemm_dbg", 0)
        t_full = timed(call)
        lib.roma_tuning(b"gemm_dbg", 256)
        t_noepi = timed(call)
        lib.roma_tuning(b"gemm_dbg", 1)
        t_nostore = timed(call)
        print(json.dumps({"shape": tag, "M": M, "N": N, "K": K, "act": act, "tiles": tiles, "workgroups": cap, "rounds": rounds,
                          "full_us": round(t_full, 1), "no_epilogue_us": round(t_noepi, 1), "no_stores_us": round(t_nostore, 1),
                          "epilogue_us_per_round": round((t_full - t_noepi) / rounds, 2),
                          "stores_us_per_round": round((t_full - t_nostore) / rounds, 2),
                          "k_tile_us": round(t_noepi / rounds / (K // 64), 3)}), flush=True)
    lib.roma_tuning(b"gemm8p_maxwg", -1)
    lib.roma_tuning(b"gemm_dbg", -1)


if __name__ == "__main__":
    torch.manual_seed(0)
    print(torch.cuda.get_device_name(0))
    shape("DINOv2 fc1, bias only", 25616, 4096, 1024, 0)
    shape("DINOv2 fc1 + GELU", 25616, 4096, 1024, 2)
    shape("DINOv2 fc2 (plain)", 25616, 1024, 4096, 0)
