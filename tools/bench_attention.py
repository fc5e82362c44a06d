"""Attention kernels at the model's shapes: time per launch, TFLOP/s and a checksum of the output bits, so that two builds
(ROMA_LIB_DIR=... selects another library directory) can be compared on one box.

    python tools/bench_attention.py
"""
import ctypes as C
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()


def P(t):
    return C.c_void_p(t.data_ptr())


def case(tag, B, heads, hd, N, iters=20):
    npad = (N + 127) // 128 * 128
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.zeros((B, heads, npad, hd), dtype=torch.bfloat16, device="cuda")
    k = torch.zeros_like(q)
    vt = torch.zeros((B, heads, hd, npad), dtype=torch.bfloat16, device="cuda")
    q[:, :, :N] = (torch.randn(B, heads, N, hd, generator=g, device="cuda") * 1.2 / hd ** 0.5).to(torch.bfloat16)
    k[:, :, :N] = (torch.randn(B, heads, N, hd, generator=g, device="cuda") * 1.5).to(torch.bfloat16)
    vt[:, :, :, :N] = torch.randn(B, heads, hd, N, generator=g, device="cuda").to(torch.bfloat16)
    o = torch.zeros((B * N, heads * hd), device="cuda", dtype=torch.bfloat16)

    def fn():
        assert lib.roma_op_attention(P(q), P(k), P(vt), P(o), B, heads, N, npad, hd, 1, 1, None) == 0, _lib.last_error(lib)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters
    ref = torch.nn.functional.scaled_dot_product_attention(q[:, :, :N].float(), k[:, :, :N].float(), vt[:, :, :, :N].float().transpose(2, 3), scale=0.6931471805599453)
    err = float((o.view(B, N, heads, hd).permute(0, 2, 1, 3).float() - ref).abs().max())
    h = hashlib.sha1(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
    print(f"{tag:30s} B={B:3d} heads={heads:3d} hd={hd:4d} N={N:5d} {t * 1e3:8.1f} us {4.0 * B * heads * N * N * hd / 1e9 / t:6.0f} TFLOP/s  max|err| vs f32 sdpa {err:.2e}  sha1 {h}", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), "lib dir:", os.environ.get("ROMA_LIB_DIR", "(in tree)"))
    lib.roma_tuning(b"attn_exp2", 1)  # the 2^x softmax the model runs (q pre-scaled by log2 e): reference scale = ln 2
    case("DINOv2 block, 16 images", 16, 16, 64, 1601)
    case("DINOv2 block, 8 images", 8, 16, 64, 1601)
    case("decoder block, 16 dpairs", 16, 8, 128, 1600)
    case("decoder block, 8 dpairs", 8, 8, 128, 1600)
