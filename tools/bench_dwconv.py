"""dwconv5x5 alone at the five wide-refiner shapes of the 560 -> 864 workload; ROMA_DW_RING=0 / 2 selects the register-prefetch
/ the ring kernel (read once per process)."""
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
    torch.manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, Cp, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.randn(25, Cp, device="cuda") * 0.1
    b = torch.randn(Cp, device="cuda") * 0.1
    for _ in range(3):
        assert lib.roma_op_dwconv5x5(P(x), P(y), P(w), P(b), B, H, W, Cp, 1, None) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        lib.roma_op_dwconv5x5(P(x), P(y), P(w), P(b), B, H, W, Cp, 1, None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    gb = 2.0 * B * H * W * Cp * 2 / 1e9
    h = hashlib.sha1(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
    print(f"ring={os.environ.get('ROMA_DW_RING', 'default'):>7s} B{B} {H}x{W} C={Cp}: {us:8.1f} us {gb / us * 1e3:6.2f} TB/s  sha1 {h}", flush=True)


if __name__ == "__main__":
    for B in (16, 8):  # 8 = one sub-batch stream of the default schedule
        for (H, Cp) in [(40, 1408), (70, 1152), (140, 576), (108, 1152), (216, 576)]:
            run(B, H, H, Cp)
