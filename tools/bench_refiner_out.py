"""out_conv + running flow / certainty update (roma_op_refiner_out) at the model's wide-scale shapes: time per launch, the
bytes it has to read per second, and a checksum of the updated flow / certainty so that two builds (ROMA_LIB_DIR=... selects
another library directory) can be compared on one box.

    python tools/bench_refiner_out.py
"""
import ctypes as C
import hashlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()
BF16 = 1


def P(t):
    return C.c_void_p(t.data_ptr())


def case(tag, M, Cp):
    g = torch.Generator(device="cuda").manual_seed(7)
    d = torch.randn(M, Cp, generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randn(3, Cp, generator=g, device="cuda") * Cp ** -0.5
    b = torch.randn(3, generator=g, device="cuda")
    flow0 = torch.randn(M, 2, generator=g, device="cuda")
    cert0 = torch.randn(M, generator=g, device="cuda")
    flow, cert = flow0.clone(), cert0.clone()

    def fn():
        assert lib.roma_op_refiner_out(P(d), Cp, BF16, P(w), P(b), P(flow), P(cert), M, Cp, C.c_float(0.25), C.c_float(0.125), None) == 0
    fn()
    torch.cuda.synchronize()
    sha = hashlib.sha1(flow.cpu().numpy().tobytes() + cert.cpu().numpy().tobytes()).hexdigest()[:12]
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    t = statistics.median(ts)
    gb = (M * Cp * 2 + M * 12 * 2) / 1e9
    print(f"{tag:34s} M={M:8d} Cp={Cp:5d} {t:8.1f} us {gb / t * 1e6:7.0f} GB/s  sha1 {sha}", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), "lib dir:", os.environ.get("ROMA_LIB_DIR", "(in tree)"))
    case("stride 4, 864 pass, 16 images", 16 * 216 * 216, 576)
    case("stride 4, 864 pass, 8 images", 8 * 216 * 216, 576)
    case("stride 4, 560 pass, 16 images", 16 * 140 * 140, 576)
    case("stride 8, 864 pass, 16 images", 16 * 108 * 108, 1152)
    case("stride 8, 560 pass, 16 images", 16 * 70 * 70, 1152)
    case("stride 16, 864 pass, 16 images", 16 * 54 * 54, 1408)
    case("stride 2 (C = 144), 16 images", 16 * 432 * 432, 144)
