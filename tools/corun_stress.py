"""Pairwise co-run determinism stress (GPU box): does kernel V (victim, stream 1) produce bit-identical output while
kernel A (aggressor, stream 2) runs concurrently?  Built to bisect the open issue of the sub-batch stream split
(profiles/r01_v20_stream_split_ab.log): in bf16 mode the sub-batch whose finest refiner overlaps the other
sub-batch's first kernels shows ~1 bf16 ulp patches in 1-5 % of the runs, never when the streams run alone.

Shapes are those of the stress case (3 pairs 112 -> 168, sub-batch of 2 pairs = 4 directed pairs at 168 x 168 for the
victims; a 1-pair sub-batch = 2 images at 112 x 112 for the aggressors' VGG head).  Usage:
    gpurun --timeout 300 -- 'timeout 250 python tools/corun_stress.py [rounds]'
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()
BF16, F32 = 1, 0
dev = "cuda"
_KEEP = []


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def S(stream):
    return C.c_void_p(stream.cuda_stream)


def rnd(*shape, std=1.0, dtype=torch.float32):
    return (torch.randn(*shape, device=dev) * std).to(dtype)


def ok(rc):
    assert rc == 0, lib.roma_last_error().decode()


# ------------------------------------------------------------------ victims: the stride-1 refiner of the upsample pass
NDP, HF = 4, 168
MF = NDP * HF * HF


def make_gemm(M, N, K, ldc, act=0):
    A, W, b = rnd(M, K, dtype=torch.bfloat16), rnd(N, K, std=K ** -0.5, dtype=torch.bfloat16), rnd(N)

    def run(stream):
        with torch.cuda.stream(stream):  # zero-fill ordered before the GEMM (pad columns stay zero)
            out = keep(torch.zeros(M, ldc, device=dev, dtype=torch.bfloat16))
        ok(lib.roma_op_gemm(P(A), K, P(W), K, P(out), ldc, M, N, K, 1, 0, 0, 0, P(b), None, None, 0, act, 1.0, BF16, BF16, S(stream)))
        return out
    return run


def keep(t):
    """Every output stays allocated until the round's torch.cuda.synchronize(): the launches go to side streams through
    raw pointers, so the caching allocator (which only knows the default stream) would otherwise hand an aggressor's
    still-being-written buffer to the next victim launch - that aliasing, not a kernel bug, made the first version of
    this tool report O(1) differences."""
    _KEEP.append(t)
    return t


def make_dwconv(B, H, W, Cp):
    x, w, b = rnd(B, H, W, Cp, dtype=torch.bfloat16), rnd(25, Cp, std=0.1), rnd(Cp, std=0.1)

    def run(stream):
        out = keep(torch.empty_like(x))
        ok(lib.roma_op_dwconv5x5(P(x), P(out), P(w), P(b), B, H, W, Cp, BF16, S(stream)))
        return out
    return run


def make_refiner_block(B, H, W, Cp):
    x, w, b = rnd(B, H, W, Cp, dtype=torch.bfloat16), rnd(25, Cp, std=0.1), rnd(Cp, std=0.1)
    pw, pb = rnd(Cp, Cp, std=0.05, dtype=torch.bfloat16), rnd(Cp)

    def run(stream):
        out = keep(torch.empty_like(x))
        ok(lib.roma_op_refiner_block(P(x), P(out), P(w), P(b), P(pw), P(pb), B, H, W, Cp, BF16, S(stream)))
        return out
    return run


def make_conv3x3(B, H, W, Cin, Cout):
    x, w, b = rnd(B, H, W, Cin, dtype=torch.bfloat16), rnd(Cout, 9 * Cin, std=0.05, dtype=torch.bfloat16), rnd(Cout)

    def run(stream):
        out = keep(torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16))
        ok(lib.roma_op_conv3x3(P(x), P(w), P(b), P(out), B, H, W, Cin, Cout, 1, BF16, S(stream)))
        return out
    return run


def make_maxpool(B, H, W, Cc):
    x = rnd(B, H, W, Cc, dtype=torch.bfloat16)

    def run(stream):
        out = keep(torch.empty(B, H // 2, W // 2, Cc, device=dev, dtype=torch.bfloat16))
        ok(lib.roma_op_maxpool2x2(P(x), P(out), B, H, W, Cc, BF16, S(stream)))
        return out
    return run


def make_conv_c3(B, H, W):
    img, w, b = rnd(B, 3, H, W), rnd(27, 64, std=0.1), rnd(64)

    def run(stream):
        out = keep(torch.empty(B, H, W, 64, device=dev, dtype=torch.bfloat16))
        ok(lib.roma_op_conv3x3_c3(P(img), P(w), P(b), P(out), B, H, W, BF16, S(stream)))
        return out
    return run


def make_chain(B, H, W, Cp, fused):
    """The refiner's nine blocks as one dependent chain (ping-pong between two buffers, exactly the product schedule):
    covers producer -> consumer hand-over between kernels of ONE stream while another stream runs (cache write-back /
    invalidate at kernel boundaries, workgroup -> XCD placement), which single-kernel victims with static inputs cannot."""
    x0 = rnd(B, H, W, Cp, dtype=torch.bfloat16)
    ws = [(rnd(25, Cp, std=0.1), rnd(Cp, std=0.1), rnd(Cp, Cp, std=Cp ** -0.5, dtype=torch.bfloat16), rnd(Cp)) for _ in range(9)]
    M = B * H * W

    def run(stream):
        with torch.cuda.stream(stream):
            a, b = x0.clone(), torch.empty_like(x0)  # produced on the victim's stream like the refiner input
        for (w, bb, pw, pb) in ws:
            if fused:
                ok(lib.roma_op_refiner_block(P(a), P(b), P(w), P(bb), P(pw), P(pb), B, H, W, Cp, BF16, S(stream)))
                a, b = b, a
            else:
                ok(lib.roma_op_dwconv5x5(P(a), P(b), P(w), P(bb), B, H, W, Cp, BF16, S(stream)))
                ok(lib.roma_op_gemm(P(b), Cp, P(pw), Cp, P(a), Cp, M, Cp, Cp, 1, 0, 0, 0, P(pb), None, None, 0, 0, 1.0, BF16, BF16, S(stream)))
        _KEEP.append((a, b))
        return a
    return run


VICTIMS = {
    "chain 9 x (dwconv + gemm) 4x168x168x24": make_chain(NDP, HF, HF, 24, False),
    "chain 9 x refiner_block  4x168x168x24": make_chain(NDP, HF, HF, 24, True),
    "gemm proj  M=112896 N=9  K=64 (ldc 24)": make_gemm(MF, 9, 64, 24),
    "gemm pw    M=112896 N=24 K=24": make_gemm(MF, 24, 24, 24),
    "gemm pw144 M=28224  N=144 K=144": make_gemm(NDP * 84 * 84, 144, 144, 144),
    "dwconv5x5  4x168x168x24": make_dwconv(NDP, HF, HF, 24),
    "refiner_block 4x168x168x24": make_refiner_block(NDP, HF, HF, 24),
    "refiner_block 4x84x84x144": make_refiner_block(NDP, 84, 84, 144),
}
AGGRESSORS = {
    "none": None,
    "conv3x3_c3 2x112x112": make_conv_c3(2, 112, 112),
    "gemm M=25088 N=64 K=32 relu": make_gemm(2 * 112 * 112, 64, 32, 64, act=1),
    "conv3x3 2x112x112 64->64": make_conv3x3(2, 112, 112, 64, 64),
    "maxpool 2x112x112x64": make_maxpool(2, 112, 112, 64),
    "conv3x3 2x56x56 128->128": make_conv3x3(2, 56, 56, 128, 128),
    "conv3x3 2x28x28 256->256": make_conv3x3(2, 28, 28, 256, 256),
}


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for vname, victim in VICTIMS.items():
        ref = victim(s1)
        torch.cuda.synchronize()
        for aname, aggr in AGGRESSORS.items():
            bad = total = 0
            worst = 0.0
            for _ in range(rounds):
                outs = []
                for _ in range(8):  # interleave so that every victim launch has aggressor launches in flight
                    if aggr is not None:
                        for _ in range(3):
                            aggr(s2)
                    outs.append(victim(s1))
                torch.cuda.synchronize()
                del _KEEP[:]
                for o in outs:
                    total += 1
                    if not torch.equal(o, ref):
                        bad += 1
                        worst = max(worst, float((o.float() - ref.float()).abs().max()))
            print(f"{vname:42s} | {aname:30s} | {bad:4d}/{total} launches differ  max|d|={worst:.3e}", flush=True)


if __name__ == "__main__":
    main()
