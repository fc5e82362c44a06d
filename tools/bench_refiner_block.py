"""refiner_block_kernel<144 | 24> alone at the benchmark's sizes; ROMA_RB_DBG (1 no depthwise phase, 2 no 1x1 phase, 4 no
output stores, 8 no ring refill) ablates its phases - one process per value (the switch is read once)."""
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


def run(B, H, W, Cp):
    torch.manual_seed(B * 1000 + H + Cp)
    x = torch.randn(B, H, W, Cp, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.randn(25, Cp, device="cuda") * 0.1
    b = torch.randn(Cp, device="cuda") * 0.1
    pw = (torch.randn(Cp, Cp, device="cuda") * 0.05).to(torch.bfloat16)
    pb = torch.randn(Cp, device="cuda")
    for _ in range(3):
        assert lib.roma_op_refiner_block(P(x), P(y), P(w), P(b), P(pw), P(pb), B, H, W, Cp, 1, None) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        lib.roma_op_refiner_block(P(x), P(y), P(w), P(b), P(pw), P(pb), B, H, W, Cp, 1, None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    gb = 2.0 * B * H * W * Cp * 2 / 1e9
    h = hashlib.sha1(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
    print(f"dbg={os.environ.get('ROMA_RB_DBG', '0'):>2s} B{B} {H}x{W} C={Cp}: {us:8.1f} us {gb / us * 1e3:6.2f} TB/s  sha1 {h}", flush=True)


if __name__ == "__main__":
    run(16, 432, 432, 144)
    run(16, 864, 864, 24)
