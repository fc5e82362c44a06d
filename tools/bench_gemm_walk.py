"""Does the ORDER in which an XCD's 32 persistent workgroups walk the output tiles matter?

gemm8p gives every XCD a contiguous band of tiles and walks it row-major (n fastest): the 32 workgroups that run side by side
cover 32 / NT tile rows x NT tile columns - at fc1 (NT = 16) 2 row panels + 16 column panels = 18 operand panels of 32 KB per
K step through the XCD's 4 MB L2.  roma_tuning("gemm8p_walk", g) walks groups of g tile rows, m fastest inside a group: the
same 32 workgroups cover g x (32 / g) tiles, g + 32 / g panels (12 at g = 4 or 8).  Results are bit-identical (only the order
of the tiles changes); the tool checks that and times each order on the DINOv2 shapes.

    python tools/bench_gemm_walk.py
"""
import ctypes as C
import hashlib
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


def timed(fn, iters=5, rounds=7):
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


def shape(tag, M, N, K, act=0, res=False):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)

    def call():
        rc = lib.roma_op_gemm(P(A), K, P(W), K, P(out), N, M, N, K, 1, 0, 0, 0, P(b), None, None, 0, act, 1.0, BF16, BF16, None)
        assert rc == 0, _lib.last_error(lib)
    row = {"shape": tag, "M": M, "N": N, "K": K, "act": act}
    ref = None
    has_knob = lib.roma_tuning(b"gemm8p_walk", 1) == 0  # a library from before the knob (ROMA_LIB_DIR=...): row-major only
    for g in (1, 2, 4, 8, 1, 4, 0, 0) if has_knob else (1, 1, 1):
        if has_knob:
            assert lib.roma_tuning(b"gemm8p_walk", g if g else -1) == 0  # 0 here = the dispatcher's own choice
        out.zero_()
        t = timed(call)
        d = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
        ref = ref or d
        assert d == ref, (tag, g, d, ref)
        row.setdefault(f"g{g}_us", []).append(round(t, 1))
    if has_knob:
        lib.roma_tuning(b"gemm8p_walk", -1)
    for key in [k for k in row if k.endswith("_us")]:
        row["TFLOPs_" + key[:-3]] = round(2.0 * M * N * K / min(row[key]) / 1e6, 1)
    row["sha1"] = ref
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    print(torch.cuda.get_device_name(0))
    shape("DINOv2 qkv (plain)", 25616, 3072, 1024)
    shape("DINOv2 fc1 + GELU", 25616, 4096, 1024, act=2)
    shape("DINOv2 fc2 (plain)", 25616, 1024, 4096)
    shape("DINOv2 proj (plain)", 25616, 1024, 1024)
    shape("8192^3", 8192, 8192, 8192)
    shape("decoder 1x1 (plain)", 78400, 1152, 1152)
