"""Operator parity on a real MI355X: every HIP kernel, called through the C ABI (ctypes), against a
plain torch-CPU fp32/fp64 evaluation of the same operator and - where the reference defines the
operator - against goldens produced by the reference itself (tests/golden/ops_reference.npz)."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

F32, BF16 = 0, 1


_KEEP = []


def P(t):
    """device pointer; keeps the tensor alive (inline `.cuda()` temporaries must outlive the async launch)"""
    if t is None:
        return None
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:-64]
    return C.c_void_p(t.data_ptr())


def rnd(*shape, seed=0, std=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(size=shape, dtype=np.float32) * np.float32(std))


@pytest.fixture(scope="module")
def lib(built_lib):
    assert torch.cuda.is_available()
    return built_lib


def ok(lib, rc):
    assert rc == 0, lib.roma_last_error().decode()


def gemm(lib, A, W, bias=None, scale=None, res=None, act=0, alpha=1.0, dt_in=F32, dt_out=F32, out=None, batch=1,
         M=None, N=None, K=None, lda=None, ldw=None, ldc=None, sA=0, sW=0, sC=0, ldr=0):
    M = M or A.shape[-2]
    K = K or A.shape[-1]
    N = N or W.shape[-2]
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), device="cuda",
                          dtype=torch.float32 if dt_out == F32 else torch.bfloat16)
    ok(lib, lib.roma_op_gemm(P(A), lda or A.stride(-2), P(W), ldw or W.stride(-2), P(out), ldc or out.stride(-2), M, N, K, batch,
                             sA, sW, sC, P(bias), P(scale), P(res), ldr, act, alpha, dt_in, dt_out, None))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("M,N,K", [(300, 200, 72), (128, 128, 32), (1000, 24, 24), (513, 64, 136), (77, 4097, 1024), (260, 1384, 1384),
                                   (4100, 576, 136), (3000, 1024, 96), (9000, 64, 72), (8200, 128, 64), (2500, 144, 144)])
def test_gemm_f32_epilogues(lib, M, N, K):
    A, W, b, s, r = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(N, seed=4), rnd(M, N, seed=5)
    ref = (A.double() @ W.double().T + b.double())
    out = gemm(lib, A.cuda(), W.cuda(), bias=b.cuda())
    assert torch.allclose(out.cpu().double(), ref, atol=2e-4 * math.sqrt(K), rtol=1e-5)
    ref2 = F.gelu(ref) * s.double() + r.double()
    out2 = gemm(lib, A.cuda(), W.cuda(), bias=b.cuda(), scale=s.cuda(), res=r.cuda(), act=2, ldr=N)
    assert torch.allclose(out2.cpu().double(), ref2, atol=2e-4 * math.sqrt(K), rtol=1e-5)
    out3 = gemm(lib, A.cuda(), W.cuda(), bias=b.cuda(), act=1)
    assert torch.allclose(out3.cpu().double(), F.relu(ref), atol=2e-4 * math.sqrt(K), rtol=1e-5)


def test_gemm_f32_is_exact_fma_chain(lib):
    # f32 MFMA == k-ordered fmaf chain (no reduced-precision path): integer-valued data must be exact
    A = torch.randint(-8, 9, (256, 96)).float()
    W = torch.randint(-8, 9, (192, 96)).float()
    out = gemm(lib, A.cuda(), W.cuda())
    assert torch.equal(out.cpu(), A @ W.T)


def test_gemm_batched_strided_accumulate(lib):
    b, M, N, K = 3, 130, 192, 64
    A, W, Cm = rnd(b, M, K, seed=1), rnd(b, N, K, seed=2), rnd(b, M, N, seed=3)
    out = Cm.clone().cuda()
    gemm(lib, A.cuda(), W.cuda(), res=out, alpha=-1.0, out=out, batch=b, M=M, N=N, K=K, sA=M * K, sW=N * K, sC=M * N, ldr=N)
    ref = Cm.double() - A.double() @ W.double().transpose(1, 2)
    assert torch.allclose(out.cpu().double(), ref, atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(300, 200, 72), (1000, 24, 24), (512, 1024, 4096), (4100, 1152, 256), (3000, 1024, 192), (9000, 64, 72), (2500, 144, 144)])
def test_gemm_bf16(lib, M, N, K):
    A, W, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2).bfloat16(), rnd(N, seed=3)
    ref = A.double() @ W.double().T + b.double()
    out = gemm(lib, A.cuda(), W.cuda(), bias=b.cuda(), dt_in=BF16, dt_out=F32)
    assert torch.allclose(out.cpu().double(), ref, atol=1e-3 * math.sqrt(K), rtol=1e-4)  # exact products, f32 accumulate
    outb = gemm(lib, A.cuda(), W.cuda(), bias=b.cuda(), dt_in=BF16, dt_out=BF16)
    assert torch.allclose(outb.cpu().double(), ref, atol=0.02 * math.sqrt(K), rtol=1e-2)


# ------------------------------------------------------------------ big-M bf16 shapes (persistent 8-wave 256 x 256 tiles)
@pytest.mark.parametrize("M,N,K", [(8192 + 37, 640, 192), (8448, 512, 64), (70000, 1024, 128), (9000, 768, 1024), (25616, 1024, 4096),
                                   (3202, 4096, 1024), (3202, 3072, 1024)])  # single-pair token rows on the 8-phase kernel (N >= 2048)
def test_gemm_big_m_bf16(lib, M, N, K):
    """The persistent 8-wave kernel with the carried last k-group: ragged last m-tile, partial n-tile (640), a single
    K slab, several tiles per persistent workgroup (70000 x 1024), long K.  Run twice (timing-dependent races)."""
    A, W, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, std=K ** -0.5).bfloat16(), rnd(N, seed=3)
    ref = A.double() @ W.double().T + b.double()
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    for _ in range(2):
        out = gemm(lib, Ad, Wd, bias=bd, dt_in=BF16, dt_out=BF16)
        assert torch.allclose(out.cpu().double(), ref, atol=0.03, rtol=1e-2)
    outg = gemm(lib, Ad, Wd, bias=bd, act=2, dt_in=BF16, dt_out=BF16)
    assert torch.allclose(outg.cpu().double(), F.gelu(ref), atol=0.03, rtol=1e-2)
    # f32 output with LayerScale and an in-place f32 residual (the ViT proj / fc2 form)
    s, r = rnd(N, seed=4), rnd(M, N, seed=5)
    x = r.clone().cuda()
    gemm(lib, Ad, Wd, bias=bd, scale=s.cuda(), res=x, out=x, dt_in=BF16, dt_out=F32, ldr=N)
    ref2 = ref * s.double() + r.double()
    assert torch.allclose(x.cpu().double(), ref2, atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("M,N,K", [(300, 64, 72), (1000, 1024, 512), (25616, 1024, 1024), (9000, 1024, 4096)])
def test_gemm_bf16_residual_in_place(lib, M, N, K):
    """x = bf16(x + ls * (A W^T + b)) with x bf16, updated in place - the residual update of the DINOv2 blocks in bf16
    mode (small 4-wave tiles, ragged last m-tile, the persistent 8-wave kernel).  The branch is rounded to bf16 before
    the add (like the reference's bf16 backbone), so the bound is one rounding of the branch plus one of the sum."""
    A, W, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, std=K ** -0.5).bfloat16(), rnd(N, seed=3)
    s, r = rnd(N, seed=4), (3.0 * rnd(M, N, seed=5)).bfloat16()
    branch = (A.double() @ W.double().T + b.double()) * s.double()
    ref = branch + r.double()
    Ad, Wd, bd, sd = A.cuda(), W.cuda(), b.cuda(), s.cuda()
    for _ in range(2):
        x = r.clone().cuda()
        ok(lib, lib.roma_op_gemm_res_bf16(P(Ad), K, P(Wd), K, P(x), N, M, N, K, P(bd), P(sd), P(x), N, None))
        torch.cuda.synchronize()
        err = (x.cpu().double() - ref).abs()
        bound = 2.0 ** -8 * (ref.abs() + branch.abs()) + 1e-3
        assert bool((err <= bound).all()), float((err - bound).max())
    # misuse is refused, not silently mis-computed: N not a multiple of 8
    x = r.clone().cuda()
    assert lib.roma_op_gemm_res_bf16(P(Ad), K, P(Wd), K, P(x), N, M, N - 4, K, P(bd), P(sd), P(x), N, None) != 0


def test_layernorm_bf16_input(lib):
    x, w, b = (rnd(1603, 1024, seed=1, std=3.0) + 0.5).bfloat16(), rnd(1024, seed=2), rnd(1024, seed=3)
    out = torch.empty((1603, 1024), device="cuda", dtype=torch.bfloat16)
    ok(lib, lib.roma_op_layernorm_dt(P(x.cuda()), BF16, P(w.cuda()), P(b.cuda()), P(out), 1603, 1024, 1e-6, BF16, None))
    torch.cuda.synchronize()
    ref = F.layer_norm(x.double(), (1024,), w.double(), b.double(), 1e-6)
    assert torch.allclose(out.cpu().double(), ref, atol=1e-3, rtol=2.0 ** -8)
    # bf16 input with f32 output is not a supported combination: refused
    o32 = torch.empty((1603, 1024), device="cuda")
    assert lib.roma_op_layernorm_dt(P(x.cuda()), BF16, P(w.cuda()), P(b.cuda()), P(o32), 1603, 1024, 1e-6, F32, None) != 0


def test_qkv_scatter_big_m_bf16(lib):
    """QKV epilogue on the persistent 8-wave kernel (rows padded to Npad per image): B*N >= 8192 rows."""
    _attention_case(lib, 6, 4, 64, 1601, BF16)
    _attention_case(lib, 5, 2, 128, 1700, BF16)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 9, 11, 64, 128), (1, 16, 16, 128, 64), (3, 56, 60, 64, 64), (3, 56, 60, 64, 128), (2, 72, 70, 64, 256)])
def test_conv3x3_implicit_gemm(lib, dt, B, H, W, Cin, Cout):
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, std=0.05), rnd(Cout, seed=3)
    tdt = torch.float32 if dt == F32 else torch.bfloat16
    xq, wq = x.to(tdt), w.to(tdt)
    ref = F.relu(F.conv2d(xq.double(), wq.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    xin = xq.permute(0, 2, 3, 1).contiguous().cuda()
    wp = wq.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()  # [cout][(ky,kx),ci]
    out = torch.empty((B, H, W, Cout), device="cuda", dtype=tdt)
    ok(lib, lib.roma_op_conv3x3(P(xin), P(wp), P(b.cuda()), P(out), B, H, W, Cin, Cout, 1, dt, None))
    torch.cuda.synchronize()
    tol = 2e-4 if dt == F32 else 3e-2
    assert torch.allclose(out.cpu().double(), ref, atol=tol, rtol=tol)


# ------------------------------------------------------------------ the 8-phase kernel (gemm8p.hip) against the classic one
@pytest.mark.parametrize("M,N,K,act,out_f32", [
    (8192, 256, 256, 0, False),      # smallest problem it takes: one n-tile, 4 K tiles (prologue + stream tail only)
    (9000, 768, 1024, 2, False),     # ragged last m-tile, 3 n-tiles, GELU
    (25616, 1024, 1024, 0, False),   # DINOv2 proj shape: 101 m-tiles, the last with 16 rows, several tiles per workgroup
    (25600, 1408, 1408, 1, False),   # stride-16 refiner 1x1: half-empty last n-tile (1408 = 5.5 x 256), ReLU
    (16384, 1024, 512, 0, True),     # f32 output
    (70000, 512, 320, 0, False),     # many tiles per persistent workgroup, K = 5 tiles
    # the 256 x 192 sibling (gemm6p.hip): 3 phases per K tile, 96-column wave tiles
    (78400, 1152, 1152, 0, False),   # stride-8 refiner 1x1: 6 n-tiles, ragged last m-tile (78400 = 306.25 x 256), ~6 tiles per workgroup
    (9000, 576, 576, 1, False),      # stride-4 refiner width, 3 n-tiles, ragged m, ReLU, fewer tiles than workgroups
    (8200, 1096, 256, 0, False),     # ragged last n-tile (1096 = 5 x 192 + 136) and 8 rows in the last m-tile, shortest K
    (16384, 576, 512, 0, True),      # f32 output
    (100000, 384, 320, 1, False),    # many tiles per persistent workgroup, K = 5 tiles (odd: both LDS buffers start a tile)
])
@pytest.mark.parametrize("sched", [1])
def test_gemm8p_matches_classic_and_reference(lib, M, N, K, act, out_f32, sched):
    """Both main loops accumulate in the same k order, so the 8-phase kernel must reproduce the classic kernel BIT FOR
    BIT (any race / mis-synchronised LDS-DMA shows up as a difference); the result is also checked against f64, and the
    8-phase launch is repeated (timing-dependent hazards).  sched: the K-loop schedule of gemm8p (1 = k-half phases; the
    quadrant-phase schedule 0 of rounds 2-3 is compiled into tools builds only since round 5) - gemm6p shapes ignore it."""
    A, W, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, std=K ** -0.5).bfloat16(), rnd(N, seed=3)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    dto = F32 if out_f32 else BF16
    try:
        lib.roma_tuning(b"gemm8p", 0)
        base = gemm(lib, Ad, Wd, bias=bd, act=act, dt_in=BF16, dt_out=dto)
        lib.roma_tuning(b"gemm8p", 1)
        lib.roma_tuning(b"gemm8p_sched", sched)
        for _ in range(3):
            out = gemm(lib, Ad, Wd, bias=bd, act=act, dt_in=BF16, dt_out=dto)
            assert torch.equal(out, base), float((out.float() - base.float()).abs().max())
    finally:
        lib.roma_tuning(b"gemm8p", -1)
        lib.roma_tuning(b"gemm8p_sched", -1)
    ref = A[:2048].double() @ W.double().T + b.double()
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    assert torch.allclose(out[:2048].cpu().double(), ref, atol=2e-3 if out_f32 else 0.03, rtol=1e-4 if out_f32 else 1e-2)


def test_gemm8p_buffer_descriptor_extents(lib):
    """Round 6: the 8-phase kernels address their operands through buffer descriptors (32-bit byte offsets below 2^31, rows
    outside M / N answered with zeros by the hardware's range check).  (a) An operand VIEW - lda > K, the activation a column
    slice of a wider matrix - must read exactly its K columns: num_records ends at the last row's last element, and what lies
    behind a row's K columns inside lda is never touched (poisoned with NaN here).  (b) An operand of 2 GiB or more cannot be
    described; the dispatcher hands it to the classic flat-address kernel, which must agree with an f64 reference."""
    M, N, K, LDA = 9000, 512, 512, 1280
    g = torch.Generator().manual_seed(11)
    wide = torch.full((M, LDA), float("nan"), dtype=torch.bfloat16)
    A = (torch.randn(M, K, generator=g)).bfloat16()
    wide[:, 256:256 + K] = A
    W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, generator=g)
    wd, Wd, bd = wide.cuda(), W.cuda(), b.cuda()
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    view = wd[:, 256:256 + K]
    ok(lib, lib.roma_op_gemm(P(view), LDA, P(Wd), K, P(out), N, M, N, K, 1, 0, 0, 0, P(bd), None, None, 0, 0, 1.0, BF16, BF16, None))
    torch.cuda.synchronize()
    ref = A.double() @ W.double().T + b.double()
    assert torch.isfinite(out.float()).all()
    assert torch.allclose(out.cpu().double(), ref, atol=0.03, rtol=1e-2)
    # (b) 2.2 GiB activation: M x K x 2 bytes >= 2^31
    M2, K2, N2 = 1_100_000, 1024, 256
    A2 = torch.randn(M2, K2, device="cuda", dtype=torch.bfloat16)
    W2 = (torch.randn(N2, K2, device="cuda") * K2 ** -0.5).to(torch.bfloat16)
    b2 = torch.randn(N2, device="cuda")
    out2 = torch.empty((M2, N2), device="cuda", dtype=torch.bfloat16)
    ok(lib, lib.roma_op_gemm(P(A2), K2, P(W2), K2, P(out2), N2, M2, N2, K2, 1, 0, 0, 0, P(b2), None, None, 0, 0, 1.0, BF16, BF16, None))
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 64), torch.arange(M2 // 2 - 32, M2 // 2 + 32), torch.arange(M2 - 64, M2)]).cuda()
    ref2 = A2[rows].double() @ W2.double().T + b2.double()
    assert torch.allclose(out2[rows].double(), ref2, atol=0.03, rtol=1e-2)


@pytest.mark.parametrize("M,N,K", [(9000, 1024, 512), (2900, 6400, 256), (256 * 37 + 5, 2048, 256)])
def test_gemm8p_walk_order_is_only_an_order(lib, M, N, K):
    """Round 6: the persistent 8-phase kernel walks each XCD's band of tiles in groups of g tile rows (roma_tuning
    "gemm8p_walk"; the dispatcher picks 8 from 24 tile columns on, row-major below).  Every g must visit every tile exactly
    once - bit-identical outputs, no NaN left from the poisoned output - including tile-row counts that are no multiple of g
    (a short last group), bands that cut through a group, and the dispatcher's own choice on a 25-column problem; f64
    reference on the row-major result."""
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, generator=g)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    outs = {}
    try:
        for gm in (1, 2, 3, 4, 8, 64, -1):
            ok(lib, lib.roma_tuning(b"gemm8p_walk", gm))
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            ok(lib, lib.roma_op_gemm(P(Ad), K, P(Wd), K, P(out), N, M, N, K, 1, 0, 0, 0, P(bd), None, None, 0, 0, 1.0, BF16, BF16, None))
            torch.cuda.synchronize()
            outs[gm] = out
    finally:
        lib.roma_tuning(b"gemm8p_walk", -1)
    assert torch.isfinite(outs[1].float()).all()
    for gm, o in outs.items():
        assert torch.equal(o.view(torch.int16), outs[1].view(torch.int16)), f"walk group {gm} differs from the row-major walk"
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)])
    ref = A[rows].double() @ W.double().T + b.double()
    assert torch.allclose(outs[1][rows.cuda()].cpu().double(), ref, atol=0.03, rtol=1e-2)


@pytest.mark.parametrize("M,act", [(65536, 0), (74656, 1), (313600, 0), (65536 + 32 * 113, 1), (746496, 1)])
def test_ws1x1_matches_tile_kernels_and_reference(lib, M, act):
    """ws1x1.hip (weight-stationary N = K = 576 refiner 1x1: W in registers, pixel chunks through a two-stage LDS-DMA ring,
    one- and two-stream workgroups) accumulates in the GEMM kernels' k order: bit-identical to gemm6p and to the classic
    kernel, repeated launches agree (ring / barrier races), f64 reference on a sample of rows.  M values: the smallest it
    takes, streams with an odd number of chunks and ragged last streams, the two model shapes."""
    import ctypes as C
    N = K = 576
    A, W, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, std=K ** -0.5).bfloat16(), rnd(N, seed=3)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    outs = {}
    try:
        for name, g8, ws in (("classic", 0, 0), ("gemm6p", 1, 0), ("ws1x1", 1, 1), ("ws1x1_again", 1, 1)):
            lib.roma_tuning(b"gemm8p", g8)
            lib.roma_tuning(b"ws1x1", ws)
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            ok(lib, lib.roma_op_gemm(P(Ad), K, P(Wd), K, P(out), N, M, N, K, 1, 0, 0, 0, P(bd), None, None, 0, act, 1.0, BF16, BF16, None))
            torch.cuda.synchronize()
            outs[name] = out
    finally:
        lib.roma_tuning(b"gemm8p", -1)
        lib.roma_tuning(b"ws1x1", -1)
    for name in ("gemm6p", "ws1x1", "ws1x1_again"):
        assert torch.equal(outs[name].view(torch.int16), outs["classic"].view(torch.int16)), (
            name, float((outs[name].float() - outs["classic"].float()).abs().max()))
    rows = torch.cat([torch.arange(0, 96), torch.arange(M // 2 - 40, M // 2 + 40), torch.arange(M - 96, M)])
    ref = A[rows].double() @ W.double().T + b.double()
    ref = F.relu(ref) if act == 1 else ref
    assert torch.allclose(outs["ws1x1"][rows.cuda()].cpu().double(), ref, atol=0.03, rtol=1e-2)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 70, 70, 256, 512), (3, 56, 60, 128, 256), (2, 72, 70, 64, 256), (1, 108, 108, 512, 512)])
@pytest.mark.parametrize("sched", [1])
def test_conv3x3_gemm8p_matches_classic_and_reference(lib, B, H, W, Cin, Cout, sched):
    """3x3 implicit GEMM on the 8-phase kernel (per-tap validity masks, zero page): bitwise vs the classic kernel and
    against torch conv2d (image borders, ragged last m-tile)."""
    x, w, b = rnd(B, Cin, H, W, seed=1).bfloat16(), rnd(Cout, Cin, 3, 3, seed=2, std=(9 * Cin) ** -0.5).bfloat16(), rnd(Cout, seed=3)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()
    bd = b.cuda()
    outs = {}
    try:
        lib.roma_tuning(b"gemm8p_sched", sched)
        for mode in (0, 1, 1):
            lib.roma_tuning(b"gemm8p", mode)
            out = torch.zeros((B, H, W, Cout), device="cuda", dtype=torch.bfloat16)
            ok(lib, lib.roma_op_conv3x3(P(xin), P(wp), P(bd), P(out), B, H, W, Cin, Cout, 1, BF16, None))
            torch.cuda.synchronize()
            if mode in outs:
                assert torch.equal(out, outs[mode])
            outs[mode] = out
    finally:
        lib.roma_tuning(b"gemm8p", -1)
        lib.roma_tuning(b"gemm8p_sched", -1)
    assert torch.equal(outs[0], outs[1]), float((outs[0].float() - outs[1].float()).abs().max())
    assert torch.allclose(outs[1].cpu().double(), ref, atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 37, 45, 128, 256), (3, 70, 70, 256, 512), (1, 108, 108, 512, 512), (2, 9, 11, 256, 256),
                                            (16, 70, 70, 256, 256), (1, 216, 216, 256, 256), (2, 17, 300, 128, 512)])
def test_conv3x3_patch_resident_matches_implicit_gemm_and_reference(lib, B, H, W, Cin, Cout):
    """conv_patch.hip (round 6): the VGG layers with Cout >= 256 with the activation patch + halo of a 64-channel slab resident
    in LDS across the nine taps.  Slab-major weight rows (roma_op_conv3x3_slab).  Bit-identical to the implicit GEMM on the same
    packing (gemm8p's conv form / the classic kernel: same products, same k order per accumulator), equal to the tap-major
    operator up to the summation order, and to torch conv2d on the same bf16 operands.  Image borders on all four sides,
    patches hanging over the right / bottom edge, images smaller than one patch, several tiles per persistent workgroup
    (16 x 70 x 70), both cout tiles (Cout = 512); every launch twice (timing-dependent hazards of the LDS-DMA pipeline)."""
    x, w, b = rnd(B, Cin, H, W, seed=1).bfloat16(), rnd(Cout, Cin, 3, 3, seed=2, std=(9 * Cin) ** -0.5).bfloat16(), rnd(Cout, seed=3)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)                                  # [cout][tap][ci]
    wslab = wt.reshape(Cout, 9, Cin // 64, 64).permute(0, 2, 1, 3).reshape(Cout, 9 * Cin).contiguous().cuda()  # [cout][slab][tap][64]
    wtap = wt.reshape(Cout, 9 * Cin).contiguous().cuda()
    bd = b.cuda()
    outs = {}
    try:
        for patch in (1, 0, 1):
            lib.roma_tuning(b"conv_patch", patch)
            out = torch.full((B, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
            ok(lib, lib.roma_op_conv3x3_slab(P(xin), P(wslab), P(bd), P(out), B, H, W, Cin, Cout, 1, BF16, None))
            torch.cuda.synchronize()
            if patch in outs:
                assert torch.equal(out.view(torch.int16), outs[patch].view(torch.int16))
            outs[patch] = out
    finally:
        lib.roma_tuning(b"conv_patch", -1)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[1].view(torch.int16), outs[0].view(torch.int16)), float((outs[1].float() - outs[0].float()).abs().max())
    assert torch.allclose(outs[1].cpu().double(), ref, atol=3e-2, rtol=3e-2)
    out_t = torch.empty_like(outs[1])
    ok(lib, lib.roma_op_conv3x3(P(xin), P(wtap), P(bd), P(out_t), B, H, W, Cin, Cout, 1, BF16, None))
    torch.cuda.synchronize()
    d = (out_t.float() - outs[1].float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(out_t.float().abs().max()) + 1e-6  # tap-major vs slab-major order: one bf16 ulp


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 9, 11, 64, 64), (1, 16, 16, 64, 128), (3, 56, 60, 64, 64), (3, 56, 60, 64, 128),
                                            (2, 131, 200, 64, 64), (2, 70, 129, 64, 128), (1, 280, 280, 64, 128), (1, 560, 560, 64, 64),
                                            (2, 9, 11, 128, 128), (3, 56, 60, 128, 128), (2, 70, 129, 128, 128), (9, 140, 131, 128, 128),
                                            (1, 432, 432, 128, 128)])
def test_conv3x3_weight_stationary(lib, B, H, W, Cin, Cout):
    """conv64.hip (VGG conv1_2 / conv2_1 / conv2_2: weights in registers, rows through a 4-slot LDS ring; Cin 128: partial sums of
    the two Cin halves exchanged through LDS) against torch conv2d and against the implicit GEMM it replaces: image borders,
    strips with a ragged last row block, x tiles hanging over the right edge (store count of the vmcnt accounting), images
    shorter than one strip, several strips per persistent workgroup (9 x 140 x 131).  Both accumulate in f32 over the same
    products; the order differs, so the two device paths agree to one bf16 rounding.  Run twice (timing-dependent races)."""
    x, w, b = rnd(B, Cin, H, W, seed=1).bfloat16(), rnd(Cout, Cin, 3, 3, seed=2, std=(9 * Cin) ** -0.5).bfloat16(), rnd(Cout, seed=3)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()
    bd = b.cuda()
    outs = []
    try:
        for mode in (0, 3, 3):
            lib.roma_tuning(b"conv64", mode)
            out = torch.full((B, H, W, Cout), -7.0, device="cuda", dtype=torch.bfloat16)
            ok(lib, lib.roma_op_conv3x3(P(xin), P(wp), P(bd), P(out), B, H, W, Cin, Cout, 1, BF16, None))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib.roma_tuning(b"conv64", -1)
    assert torch.equal(outs[1], outs[2])
    assert torch.allclose(outs[1].cpu().double(), ref, atol=3e-2, rtol=3e-2)
    d = (outs[1].float() - outs[0].float()).abs()
    assert float((d / (outs[0].float().abs() + 1.0)).max()) <= 2.0 ** -7, float(d.max())


def _attention_case(lib, B, heads, hd, N, dt):
    D = heads * hd
    tdt = torch.float32 if dt == F32 else torch.bfloat16
    x, w, b = rnd(B * N, D, seed=1).to(tdt), rnd(3 * D, D, seed=2, std=1.5 / math.sqrt(D)).to(tdt), rnd(3 * D, seed=3, std=0.1)
    npad = (N + 127) // 128 * 128
    q = torch.zeros((B, heads, npad, hd), device="cuda", dtype=tdt)
    k = torch.zeros_like(q)
    vt = torch.zeros((B, heads, hd, npad), device="cuda", dtype=tdt)
    ok(lib, lib.roma_op_qkv_scatter_gemm(P(x.cuda()), P(w.cuda()), P(b.cuda()), P(q), P(k), P(vt), B, N, npad, heads, hd, D, dt, dt, None))
    out = torch.empty((B * N, D), device="cuda", dtype=tdt)
    ok(lib, lib.roma_op_attention(P(q), P(k), P(vt), P(out), B, heads, N, npad, hd, dt, dt, None))
    torch.cuda.synchronize()
    qkv = (x.double() @ w.double().T + b.double()).reshape(B, N, 3, heads, hd)
    qq, kk, vv = [t.transpose(1, 2) for t in torch.unbind(qkv, 2)]
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B * N, D)
    # scatter layout check
    assert torch.allclose(k.cpu().double()[:, :, :N], kk, atol=1e-4 if dt == F32 else 3e-2)
    assert torch.allclose(vt.cpu().double()[:, :, :, :N], vv.transpose(2, 3), atol=1e-4 if dt == F32 else 3e-2)
    tol = 2e-5 if dt == F32 else 6e-2  # bf16: P and V are quantised to bf16 before the second MFMA
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < tol, err


@pytest.mark.parametrize("B,heads,hd,N", [(2, 16, 64, 65), (1, 16, 64, 257), (2, 8, 128, 64), (1, 8, 128, 200)])
def test_attention_f32(lib, B, heads, hd, N):
    _attention_case(lib, B, heads, hd, N, F32)


@pytest.mark.parametrize("B,heads,hd,N", [(2, 16, 64, 65), (1, 16, 64, 257), (2, 8, 128, 64), (1, 8, 128, 200), (1, 3, 64, 40),
                                          (2, 2, 64, 128), (1, 2, 64, 1601)])
def test_attention_bf16(lib, B, heads, hd, N):
    _attention_case(lib, B, heads, hd, N, BF16)


def _attention_direct(lib, q, k, v, N, pad_garbage=False):
    """roma_op_attention on hand-built operands: q [B,h,N,hd] (already scaled by 1/sqrt(hd)), k, v [B,h,N,hd]; bf16.
    pad_garbage: rows N .. Npad of q / k and the matching V^T columns hold large finite numbers instead of zeros."""
    B, heads, _, hd = q.shape
    npad = (N + 127) // 128 * 128
    if pad_garbage:
        g = torch.Generator().manual_seed(99)
        qd = (torch.randn(B, heads, npad, hd, generator=g) * 40.0).to(torch.bfloat16).cuda()
        kd = (torch.randn(B, heads, npad, hd, generator=g) * 40.0).to(torch.bfloat16).cuda()
        vtd = (torch.randn(B, heads, hd, npad, generator=g) * 40.0).to(torch.bfloat16).cuda()
    else:
        qd = torch.zeros((B, heads, npad, hd), device="cuda", dtype=torch.bfloat16)
        kd = torch.zeros_like(qd)
        vtd = torch.zeros((B, heads, hd, npad), device="cuda", dtype=torch.bfloat16)
    qd[:, :, :N], kd[:, :, :N] = q.cuda(), k.cuda()
    vtd[:, :, :, :N] = v.transpose(2, 3).cuda()
    out = torch.empty((B * N, heads * hd), device="cuda", dtype=torch.bfloat16)
    ok(lib, lib.roma_op_attention(P(qd), P(kd), P(vtd), P(out), B, heads, N, npad, hd, BF16, BF16, None))
    torch.cuda.synchronize()
    return out.cpu().double().reshape(B, N, heads, hd).transpose(1, 2)


@pytest.mark.parametrize("hd,N", [(64, 1601), (128, 1600), (64, 40)])
def test_attention_deferred_rescale(lib, hd, N):
    """attn_h16_v2_kernel moves its softmax reference only on the first tile and when a tile's P = e^(s - m_ref) sum to
    more than 2^14 for some query (rounds 3-5: when a score exceeded the reference by 2^8).  Random data never takes that
    branch after tile 0, so force it: for a third of the queries one LATE key (and for another third one key of the first
    tile, so that the reference starts far above everything that follows) scores >> all others - by more than e^12 >
    2^14, and for some queries by more than e^89 > 2^128, where the first attempt's 2^x overflows to +inf and the row sum
    is not finite.  Checked against an f64 softmax of the same bf16 operands (round 3 also compared with the
    always-rescaling round-1 kernel: 3.1e-2 both; that kernel is gone)."""
    B, heads = 2, 3
    g = torch.Generator().manual_seed(5)
    q = (torch.randn(B, heads, N, hd, generator=g) / math.sqrt(hd)).to(torch.bfloat16)
    k = torch.randn(B, heads, N, hd, generator=g).to(torch.bfloat16)
    v = torch.randn(B, heads, N, hd, generator=g).to(torch.bfloat16)
    late, early = N - 7, 3
    qn = q.float() / q.float().norm(dim=-1, keepdim=True)
    kf = k.float()
    sel_late = torch.arange(N) % 3 == 0
    sel_early = torch.arange(N) % 3 == 1
    # one key per (b, head) can only align with one direction: use the mean direction of the selected queries, scaled up
    kf[:, :, late] = 40.0 * math.sqrt(hd) * qn[:, :, sel_late].mean(dim=2) / qn[:, :, sel_late].mean(dim=2).norm(dim=-1, keepdim=True)
    kf[:, :, early] = 40.0 * math.sqrt(hd) * qn[:, :, sel_early].mean(dim=2) / qn[:, :, sel_early].mean(dim=2).norm(dim=-1, keepdim=True)
    k = kf.to(torch.bfloat16)
    sc = q.double() @ k.double().transpose(2, 3)
    # the construction must really cross the threshold (2^14 = e^9.7) at the late tile for some queries, in both directions
    if N > 64:
        jump = sc[:, :, :, late] - sc[:, :, :, : (late // 64) * 64].amax(dim=-1)
        assert float(jump.max()) > 12.0 and float((-jump).max()) > 12.0, (float(jump.max()), float(jump.min()))
    ref = torch.softmax(sc, dim=-1) @ v.double()
    o2 = _attention_direct(lib, q, k, v, N)
    e2 = float((o2 - ref).abs().max())
    assert e2 < 3e-2, e2  # |O| <= max |v| ~ 4; bf16 P and V
    assert torch.isfinite(o2).all()


@pytest.mark.parametrize("hd,N", [(64, 1601), (64, 65), (128, 1600), (64, 200)])
def test_attention_ignores_padding_rows(lib, hd, N):
    """Rows N .. Npad of the q / k / V^T workspaces are not part of the problem: in the model they hold whatever the other
    attention layout (DINOv2 vs decoder transformer share the workspace) left there.  The valid outputs must not depend on
    them BIT FOR BIT - the kernel takes a wave-wide decision (deferred rescale) in which padding queries must have no vote
    (they had one: run-to-run differences of the last patch token at 560 -> 864, profiles/r03_v24_attention_padding.log)."""
    B, heads = 2, 4
    g = torch.Generator().manual_seed(11)
    q = (torch.randn(B, heads, N, hd, generator=g) * 2.0 / math.sqrt(hd)).to(torch.bfloat16)
    k = (torch.randn(B, heads, N, hd, generator=g) * 2.0).to(torch.bfloat16)
    v = torch.randn(B, heads, N, hd, generator=g).to(torch.bfloat16)
    clean = _attention_direct(lib, q, k, v, N)
    dirty = _attention_direct(lib, q, k, v, N, pad_garbage=True)
    assert torch.isfinite(dirty).all()
    assert torch.equal(clean, dirty), float((clean - dirty).abs().max())


def test_layernorm(lib):
    x, w, b = rnd(37, 1024, seed=1, std=3.0) + 0.5, rnd(1024, seed=2), rnd(1024, seed=3)
    out = torch.empty((37, 1024), device="cuda")
    ok(lib, lib.roma_op_layernorm(P(x.cuda()), P(w.cuda()), P(b.cuda()), P(out), 37, 1024, 1e-6, F32, None))
    torch.cuda.synchronize()
    ref = F.layer_norm(x.double(), (1024,), w.double(), b.double(), 1e-6)
    assert torch.allclose(out.cpu().double(), ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("n,d,batch", [(64, 512, 2), (256, 512, 3), (1600, 512, 1)])
def test_cholesky_solve(lib, n, d, batch):
    # SPD matrices shaped like the GP system: exp((cos-1)/0.2) Gram + 0.1 I   (matcher.py:191-200, 301)
    y = rnd(batch, n, 32, seed=1)
    yn = y / y.norm(dim=-1, keepdim=True)
    A = torch.exp((yn @ yn.transpose(1, 2) - 1.0) / 0.2) + 0.1 * torch.eye(n)
    Fm = rnd(batch, n, d, seed=2)
    ref = torch.cholesky_solve(Fm.double(), torch.linalg.cholesky(A.double()))
    Ad = A.clone().cuda()
    Ft = Fm.transpose(1, 2).contiguous().cuda()
    LT = torch.empty_like(Ad)
    Linv = torch.empty((batch, n // 64, 64, 64), device="cuda")
    LinvT = torch.empty_like(Linv)
    ok(lib, lib.roma_op_cholesky_solve_t(P(Ad), P(Ft), P(LT), P(Linv), P(LinvT), n, d, batch, None))
    torch.cuda.synchronize()
    X = Ft.cpu().transpose(1, 2).double()
    assert torch.allclose(X, ref, atol=2e-4, rtol=1e-4), (X - ref).abs().max()
    Lref = torch.linalg.cholesky(A.double())
    assert torch.allclose(torch.tril(Ad.cpu().double()), Lref, atol=1e-4)


@pytest.mark.parametrize("n,d", [(64, 512), (320, 512), (1600, 512)])
def test_cholesky_solve_augmented_storage_forms_agree(lib, n, d):
    """F^T stored right behind A (one (n + d) x n matrix, what gp_posterior allocates): the forward substitution is part of the
    factorisation.  Three forms of the same solve:
      * separate buffers: right-looking chain, separate forward loop (rounds 1-3);
      * augmented, roma_tuning("gp_col", 0): the same right-looking operations on every element in the same order (round 4) -
        bit-identical to the separate form;
      * augmented, default (round 6): the LEFT-looking block-column kernel (chol_col.hip), one launch per 64 columns - another
        summation order (one fmaf chain over the earlier blocks instead of in-memory subtractions), so it agrees to f32
        rounding and is held to the f64 solution like the others."""
    y = rnd(1, n, 48, seed=11)
    yn = y / y.norm(dim=-1, keepdim=True)
    A = (torch.exp((yn @ yn.transpose(1, 2) - 1.0) / 0.2) + 0.1 * torch.eye(n))[0]
    Ft = rnd(d, n, seed=12)
    outs = []
    for aug, col, leader in ((False, 0, 1), (True, 0, 1), (True, 1, 1), (True, 1, 1), (True, 1, 0)):
        ok(lib, lib.roma_tuning(b"gp_col", col))
        ok(lib, lib.roma_tuning(b"gp_col_leader", leader))  # 0: every workgroup factorises its own copy of the block (fallback form)
        buf = torch.empty(((n + d) * n + 4096,), device="cuda")
        Ad = buf[:n * n].view(n, n)
        Ad.copy_(A)
        Fd = buf[n * n:(n + d) * n].view(d, n) if aug else torch.empty((d, n), device="cuda")
        Fd.copy_(Ft)
        LT = torch.full((n, n), float("nan"), device="cuda")
        Linv = torch.empty((n // 64, 64, 64), device="cuda")
        LinvT = torch.empty_like(Linv)
        ok(lib, lib.roma_op_cholesky_solve_t(P(Ad), P(Fd), P(LT), P(Linv), P(LinvT), n, d, 1, None))
        torch.cuda.synchronize()
        outs.append((Fd.clone(), torch.tril(Ad).clone(), Linv.clone(), LinvT.clone()))
    ok(lib, lib.roma_tuning(b"gp_col", -1))
    ok(lib, lib.roma_tuning(b"gp_col_leader", -1))
    for a, b in zip(outs[0], outs[1]):   # right-looking: separate == augmented, bit for bit
        assert torch.equal(a, b)
    for a, b in zip(outs[2], outs[3]):   # the block-column kernel is deterministic
        assert torch.equal(a, b)
    for a, b in zip(outs[2], outs[4]):   # ... and the in-launch hand-off changes who computes, not what
        assert torch.equal(a, b)
    ref = torch.cholesky_solve(Ft.t().double(), torch.linalg.cholesky(A.double()))
    Lref = torch.linalg.cholesky(A.double())
    for o in (outs[1], outs[2]):
        assert torch.allclose(o[0].cpu().t().double(), ref, atol=2e-4, rtol=1e-4), float((o[0].cpu().t().double() - ref).abs().max())
        assert torch.allclose(o[1].cpu().double(), Lref, atol=1e-4), float((o[1].cpu().double() - Lref).abs().max())
        assert torch.equal(o[2].transpose(1, 2), o[3])  # Linv^T table = transpose of the Linv table
    # left- vs right-looking: the same numbers up to f32 rounding of a well-conditioned system (cond ~ 1e3)
    for a, b, tol in zip(outs[1][:3], outs[2][:3], (5e-5, 2e-5, 2e-4)):
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), float((a - b).abs().max())
    # the block inverses really invert the factor's diagonal blocks
    Lc = outs[2][1].cpu().double()
    for k in range(n // 64):
        blk = Lc[64 * k:64 * k + 64, 64 * k:64 * k + 64]
        assert torch.allclose(outs[2][2][k].cpu().double() @ blk, torch.eye(64, dtype=torch.float64), atol=1e-4)


@pytest.mark.parametrize("batch", [3])
def test_cholesky_block_column_batched_matches_f64(lib, batch):
    """chol_col.hip on a batch with the GP's strides (n = 320, d = 512): every image against the f64 solve; workgroup 0 of a
    launch must not publish the factor block where its neighbours still read the original one (the diagonal blocks of A are
    restored behind the last column)."""
    n, d = 320, 512
    y = rnd(batch, n, 40, seed=21)
    yn = y / y.norm(dim=-1, keepdim=True)
    A = torch.exp((yn @ yn.transpose(1, 2) - 1.0) / 0.2) + 0.1 * torch.eye(n)
    Ft = rnd(batch, d, n, seed=22)
    buf = torch.empty((batch, (n + d) * n), device="cuda")
    LT = torch.empty((batch, n, n), device="cuda")
    Linv = torch.empty((batch, n // 64, 64, 64), device="cuda")
    LinvT = torch.empty_like(Linv)
    for _ in range(2):  # twice from the same inputs: no state is carried between calls
        buf[:, :n * n] = A.reshape(batch, -1).cuda()
        buf[:, n * n:] = Ft.reshape(batch, -1).cuda()
        ok(lib, lib.roma_op_cholesky_solve_t(P(buf), C.c_void_p(buf.data_ptr() + n * n * 4), P(LT), P(Linv), P(LinvT), n, d, batch, None))
        torch.cuda.synchronize()
        X = buf[:, n * n:].reshape(batch, d, n).cpu().transpose(1, 2).double()
        ref = torch.cholesky_solve(Ft.transpose(1, 2).double(), torch.linalg.cholesky(A.double()))
        assert torch.allclose(X, ref, atol=2e-4, rtol=1e-4), float((X - ref).abs().max())
        Lgot = torch.tril(buf[:, :n * n].reshape(batch, n, n).cpu().double())
        assert torch.allclose(Lgot, torch.linalg.cholesky(A.double()), atol=1e-4)


def test_cls_to_flow_reference_golden(lib):
    g = np.load(os.path.join(GOLDEN, "ops_reference.npz"))
    cls = torch.from_numpy(g["c2f_cls"])  # [B,4096,H,W]
    B, Cn, H, W = cls.shape
    logits = torch.zeros((B * H * W, 4104))
    logits[:, :4096] = cls.permute(0, 2, 3, 1).reshape(-1, 4096)
    logits[:, 4096] = 7.0
    flow = torch.empty((B * H * W, 2), device="cuda")
    cert = torch.empty((B * H * W,), device="cuda")
    ok(lib, lib.roma_op_cls_to_flow(P(logits.cuda()), 4104, P(flow), P(cert), B * H * W, None))
    torch.cuda.synchronize()
    ref = torch.from_numpy(g["c2f_flow"]).reshape(-1, 2)
    assert torch.allclose(flow.cpu(), ref, atol=1e-5)
    assert torch.all(cert.cpu() == 7.0)


@pytest.mark.parametrize("name,r", [("lc_r7", 7), ("lc_r3", 3), ("lc_r2", 2)])
def test_local_corr_reference_golden(lib, name, r):
    """HIP fused kernel vs the reference's own local_correlation (torch fallback) outputs."""
    from roma_amd.local_correlation import local_corr, local_correlation
    g = np.load(os.path.join(GOLDEN, "ops_reference.npz"))
    f0, f1, warp, ref = [torch.from_numpy(g[f"{name}_{k}"]) for k in ("f0", "f1", "warp", "corr")]
    out = local_correlation(f0.cuda(), f1.cuda(), r, warp.cuda(), use_custom_corr=True)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert torch.allclose(out.cpu(), ref, atol=5e-5, rtol=1e-5), (out.cpu() - ref).abs().max()
    # plugin-signature form (arbitrary per-tap coordinates), as local_corr_wrapper builds them (local_correlation.py:24-32)
    B, c, h, w = f0.shape
    K = (2 * r + 1) ** 2
    lw = torch.meshgrid(torch.linspace(-2 * r / h, 2 * r / h, 2 * r + 1), torch.linspace(-2 * r / w, 2 * r / w, 2 * r + 1), indexing="ij")
    lw = torch.stack((lw[1], lw[0]), dim=-1).reshape(1, K, 2)
    coords = (warp.permute(0, 2, 3, 1)[..., None, :] + lw[:, None, None]).reshape(B, h * w, K, 2)
    out2 = local_corr(f0.reshape(B, c, h * w).permute(0, 2, 1).contiguous().cuda() / (c ** 0.5),
                      f1.permute(0, 2, 3, 1).contiguous().cuda(), coords.cuda())
    torch.cuda.synchronize()
    out2 = out2.permute(0, 2, 1).reshape(B, K, h, w)
    assert torch.allclose(out2.cpu(), ref, atol=5e-5, rtol=1e-5), (out2.cpu() - ref).abs().max()


@pytest.mark.parametrize("name,r", [("nn_r3", 3), ("nn_r2", 2)])
def test_local_corr_nearest_reference_golden(lib, name, r):
    """mode="nearest" of the plugin (local_correlation.py:19,30,85) against the reference's own fallback: exact half-pixel
    ties (nearbyint, to even), exact centres, out-of-range taps; window form and plugin-signature form."""
    from roma_amd.local_correlation import local_corr, local_correlation
    g = np.load(os.path.join(GOLDEN, "ops_nearest_reference.npz"))
    f0, f1, warp, ref = [torch.from_numpy(g[f"{name}_{k}"]) for k in ("f0", "f1", "warp", "corr")]
    out = local_correlation(f0.cuda(), f1.cuda(), r, warp.cuda(), sample_mode="nearest")
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert torch.allclose(out.cpu(), ref, atol=5e-5, rtol=1e-5), (out.cpu() - ref).abs().max()
    B, c, h, w = f0.shape
    K = (2 * r + 1) ** 2
    lw = torch.meshgrid(torch.linspace(-2 * r / h, 2 * r / h, 2 * r + 1), torch.linspace(-2 * r / w, 2 * r / w, 2 * r + 1), indexing="ij")
    lw = torch.stack((lw[1], lw[0]), dim=-1).reshape(1, K, 2)
    coords = (warp.permute(0, 2, 3, 1)[..., None, :] + lw[:, None, None]).reshape(B, h * w, K, 2)
    out2 = local_corr(f0.reshape(B, c, h * w).permute(0, 2, 1).contiguous().cuda() / (c ** 0.5),
                      f1.permute(0, 2, 3, 1).contiguous().cuda(), coords.cuda(), mode="nearest")
    out2 = out2.permute(0, 2, 1).reshape(B, K, h, w)
    assert torch.allclose(out2.cpu(), ref, atol=5e-5, rtol=1e-5), (out2.cpu() - ref).abs().max()
    with pytest.raises(ValueError):
        local_corr(f0.reshape(B, c, h * w).permute(0, 2, 1).contiguous().cuda(), f1.permute(0, 2, 3, 1).contiguous().cuda(),
                   coords.cuda(), mode="bicubic")


def test_local_corr_real_shape_vs_oracle(lib):
    """stride-8 shape of the real model (C=512, r=3) with a smooth + noisy warp, bf16 and f32."""
    from oracle import roma_oracle
    from roma_amd.local_correlation import local_correlation
    B, c, h, w, r = 1, 512, 35, 35, 3
    f0, f1 = rnd(B, c, h, w, seed=1), rnd(B, c, h, w, seed=2)
    warp = roma_oracle.pixel_grid(B, h, w) * 0.9 + rnd(B, 2, h, w, seed=3, std=0.02)
    ref = roma_oracle.local_correlation(f0, f1, r, warp)
    out = local_correlation(f0.cuda(), f1.cuda(), r, warp.cuda())
    assert torch.allclose(out.cpu(), ref, atol=1e-4, rtol=1e-5), (out.cpu() - ref).abs().max()
    outb = local_correlation(f0.cuda().bfloat16(), f1.cuda().bfloat16(), r, warp.cuda())
    refb = roma_oracle.local_correlation(f0.bfloat16().float(), f1.bfloat16().float(), r, warp)
    assert torch.allclose(outb.cpu(), refb, atol=1e-3, rtol=1e-4), (outb.cpu() - refb).abs().max()


@pytest.mark.parametrize("r,c,h,w", [(2, 256, 37, 43), (3, 512, 35, 35), (7, 512, 20, 24)])
def test_local_corr_tiled_gather_and_legacy_paths_agree(lib, r, c, h, w):
    """The three forms of the window kernel - LDS-staged tiles (coherent warps), the per-tile gather work list
    (incoherent warps) and the per-pixel kernel of round 1 - against the oracle and each other, on a warp that mixes
    both regimes: a smooth zooming / shifting field (tile path, incl. windows that leave the image on two sides and
    sizes that are not multiples of the 8 x 8 tile), a noisy band (work list) and far-outside targets."""
    from oracle import roma_oracle
    from roma_amd.local_correlation import local_correlation
    B = 2
    f0, f1 = rnd(B, c, h, w, seed=1), rnd(B, c, h, w, seed=2)
    warp = roma_oracle.pixel_grid(B, h, w) * 1.15 + 0.08 + rnd(B, 2, h, w, seed=3, std=0.004)
    warp[:, :, h // 2:h // 2 + 5] += rnd(B, 2, 5, w, seed=4, std=0.5)      # incoherent band
    warp[1, :, :3, :3] = 3.0                                                  # far outside: all taps zero
    ref = roma_oracle.local_correlation(f0, f1, r, warp)
    refb = roma_oracle.local_correlation(f0.bfloat16().float(), f1.bfloat16().float(), r, warp)
    outs = {}
    try:
        # lc_mode 0: tiles (16-bit: all-pairs on the matrix core; f32: VALU dots) + the incoherent tiles' queries, 1: every tile
        # treated as incoherent, 2: per pixel;  lc_bin 1 (round 6): incoherent queries sorted by target bin and served by the
        # LIST form of the tile kernel, 0: per-query gathers (rounds 2-5)
        for mode, binned in ((0, 1), (1, 1), (0, 0), (1, 0), (2, 0)):
            lib.roma_tuning(b"lc_mode", mode)
            lib.roma_tuning(b"lc_bin", binned)
            out = local_correlation(f0.cuda(), f1.cuda(), r, warp.cuda())
            assert torch.allclose(out.cpu(), ref, atol=1e-4, rtol=1e-5), (mode, binned, float((out.cpu() - ref).abs().max()))
            outb = local_correlation(f0.cuda().bfloat16(), f1.cuda().bfloat16(), r, warp.cuda())
            assert torch.allclose(outb.cpu(), refb, atol=1e-3, rtol=1e-4), (mode, binned, float((outb.cpu() - refb).abs().max()))
            outs[(mode, binned)] = (out.cpu(), outb.cpu())
        # the binned form is deterministic although the order of the queries inside a bin is not (atomics): run it again
        lib.roma_tuning(b"lc_mode", 1)
        lib.roma_tuning(b"lc_bin", 1)
        for _ in range(3):
            assert torch.equal(local_correlation(f0.cuda().bfloat16(), f1.cuda().bfloat16(), r, warp.cuda()).cpu(), outs[(1, 1)][1])
            assert torch.equal(local_correlation(f0.cuda(), f1.cuda(), r, warp.cuda()).cpu(), outs[(1, 1)][0])
    finally:
        lib.roma_tuning(b"lc_mode", -1)
        lib.roma_tuning(b"lc_bin", -1)
    # the gather-list and legacy forms run the same per-pixel code: bit-identical; the tiled forms sum in another order
    assert torch.equal(outs[(1, 0)][0], outs[(2, 0)][0]) and torch.equal(outs[(1, 0)][1], outs[(2, 0)][1])
    assert torch.allclose(outs[(0, 0)][0], outs[(1, 0)][0], atol=2e-5, rtol=1e-5)
    # tile form and LIST form evaluate every (window pixel, query) dot product with the same instruction sequence: a query's
    # result does not depend on which of the two served it
    assert torch.equal(outs[(0, 1)][1], outs[(1, 1)][1])
    assert torch.allclose(outs[(0, 1)][0], outs[(1, 1)][0], atol=2e-5, rtol=1e-5)
    assert torch.allclose(outs[(0, 0)][1], outs[(1, 0)][1], atol=2e-5, rtol=1e-5)   # 16-bit: MFMA all-pairs tiles vs per-query dots (exact products, f32 sums)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("C,E,K,ldf,ldd,h,w", [(512, 64, 49, 512, 1152, 21, 19),   # stride 8: vector kernel, 64 lanes per pixel
                                               (256, 32, 25, 256, 576, 30, 28),    # stride 4
                                               (64, 16, 0, 64, 144, 33, 40),       # stride 2: several pixels per wave
                                               (9, 6, 0, 16, 24, 50, 47)])         # stride 1: the pix<9,6> kernel
def test_refiner_input_grid_sample_warp(lib, dt, C, E, K, ldf, ldd, h, w):
    """roma_op_refiner_input (the F.grid_sample warp of matcher.py:132-134 fused with the concat writer and disp_emb) in
    isolation against torch float64: x copy, bilinear zero-padded sample of the support image (incl. out-of-range and
    exact-centre coordinates), displacement embedding, zero padding, correlation slice left untouched."""
    B, nimg, shift = 2, 4, 2
    tdt = torch.float32 if dt == F32 else torch.bfloat16
    feat = torch.zeros(nimg, h * w, ldf)
    feat[:, :, :C] = rnd(nimg, h * w, C, seed=1)
    feat = feat.to(tdt)
    ys, xs = torch.linspace(-1 + 1 / h, 1 - 1 / h, h), torch.linspace(-1 + 1 / w, 1 - 1 / w, w)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    grid = torch.stack((gx, gy), dim=-1)[None].expand(B, h, w, 2)
    flow = (grid * 1.1 + rnd(B, h, w, 2, seed=2, std=0.2)).contiguous()
    flow[0, 0, 0] = torch.tensor([-1.9, 0.3])                          # far outside: zeros
    flow[0, 0, 1] = torch.tensor([1 - 1 / w, -1 + 1 / h])              # exact pixel centre
    flow[1, 1, 2] = torch.tensor([1.0, 1.0])                           # on the border: partly outside
    emb_w, emb_b = rnd(E, 2, seed=3, std=0.7), rnd(E, seed=4, std=0.1)
    disp_scale = 1.25 * 1.5428
    d = torch.full((B, h * w, ldd), 7.0).to(tdt).cuda()                # sentinel: the correlation slice must survive
    ok(lib, lib.roma_op_refiner_input(P(feat.cuda()), ldf, P(flow.cuda()), P(d), ldd, P(emb_w.cuda()), P(emb_b.cuda()), B, h, w, C,
                                      E, K, nimg, shift, disp_scale, dt, None))
    torch.cuda.synchronize()
    got = d.cpu().double()
    f64 = feat.double()
    x = f64[:B, :, :C]
    ysrc = f64[[(b + shift) % nimg for b in range(B)], :, :C].reshape(B, h, w, C).permute(0, 3, 1, 2)
    xhat = F.grid_sample(ysrc, flow.double(), mode="bilinear", padding_mode="zeros", align_corners=False).permute(0, 2, 3, 1).reshape(B, h * w, C)
    emb = (disp_scale * (flow.double() - grid.double())).reshape(B, h * w, 2) @ emb_w.double().T + emb_b.double()
    tol = 2e-5 if dt == F32 else 2e-2
    assert torch.allclose(got[:, :, :C], x, atol=0, rtol=0)
    assert torch.allclose(got[:, :, C:2 * C], xhat, atol=tol, rtol=tol), float((got[:, :, C:2 * C] - xhat).abs().max())
    assert torch.allclose(got[:, :, 2 * C:2 * C + E], emb, atol=tol, rtol=tol)
    assert bool((got[:, :, 2 * C + E:2 * C + E + K] == 7.0).all())
    assert bool((got[:, :, 2 * C + E + K:] == 0.0).all())


@pytest.mark.parametrize("b,h,w", [(2, 8, 8), (1, 12, 10)])
def test_gp_posterior_vs_oracle(lib, b, h, w):
    """roma_op_gp = GP.forward (matcher.py:291-323) in isolation against the CPU oracle (which is pinned on the
    reference's gp16 stage by tests/test_cpu_oracle.py): cosine-kernel Gram matrices, blocked Cholesky, posterior mean."""
    from oracle import roma_oracle
    x, y = rnd(b, 512, h, w, seed=1), rnd(b, 512, h, w, seed=2)
    sd = {"decoder.gps.16.pos_conv.weight": rnd(512, 2, 1, 1, seed=3, std=0.5), "decoder.gps.16.pos_conv.bias": rnd(512, seed=4, std=0.5)}
    ref = roma_oracle.gp_posterior(x, y, sd)                                   # [b, 512, h, w]
    xt = x.permute(0, 2, 3, 1).reshape(b, h * w, 512).contiguous().cuda()
    yt = y.permute(0, 2, 3, 1).reshape(b, h * w, 512).contiguous().cuda()
    mu = torch.empty(b, h * w, 512, device="cuda")
    ok(lib, lib.roma_op_gp(P(xt), P(yt), P(sd["decoder.gps.16.pos_conv.weight"].reshape(512, 2).contiguous().cuda()),
                           P(sd["decoder.gps.16.pos_conv.bias"].cuda()), P(mu), b, h, w, F32, None))
    torch.cuda.synchronize()
    got = mu.cpu().reshape(b, h, w, 512).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=2e-4, rtol=1e-4), float((got - ref).abs().max())


@pytest.mark.parametrize("hin,hout,nc", [(40, 70, 2), (70, 140, 1), (560, 108, 2), (8, 14, 1)])
def test_resize_bilinear(lib, hin, hout, nc):
    x = rnd(2, nc, hin, hin, seed=1)
    ref = F.interpolate(x, size=(hout, hout), mode="bilinear", align_corners=False)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.empty((2, hout, hout, nc), device="cuda")
    ok(lib, lib.roma_op_resize_bilinear(P(xin), P(out), 2, hin, hin, hout, hout, nc, None))
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu().permute(0, 3, 1, 2), ref, atol=2e-5), (out.cpu().permute(0, 3, 1, 2) - ref).abs().max()


@pytest.mark.parametrize("dt,B,H,W,Cp", [(F32, 2, 13, 10, 24), (BF16, 2, 13, 10, 24), (BF16, 1, 41, 59, 264), (BF16, 1, 262, 31, 152),
                                         (BF16, 1, 20, 300, 576), (F32, 1, 20, 37, 264)])
def test_dwconv5x5(lib, dt, B, H, W, Cp):
    """Rolling-window depthwise kernel: ragged channel chunks (264 = 2 x 33 groups), x tiles, both strip heights."""
    tdt = torch.float32 if dt == F32 else torch.bfloat16
    x, w, b = rnd(B, Cp, H, W, seed=1).to(tdt), rnd(Cp, 1, 5, 5, seed=2, std=0.2), rnd(Cp, seed=3)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=2, groups=Cp)).permute(0, 2, 3, 1)
    out = torch.full((B, H, W, Cp), float("nan"), device="cuda", dtype=tdt)
    wp = w.reshape(Cp, 25).T.contiguous().cuda()
    ok(lib, lib.roma_op_dwconv5x5(P(x.permute(0, 2, 3, 1).contiguous().cuda()), P(out), P(wp), P(b.cuda()), B, H, W, Cp, dt, None))
    torch.cuda.synchronize()
    tol = 1e-5 if dt == F32 else 2e-2
    assert torch.allclose(out.cpu().double(), ref, atol=tol, rtol=tol)


@pytest.mark.parametrize("Cp,B,H,W", [(576, 2, 40, 36), (576, 1, 3, 50), (1152, 2, 27, 30), (1408, 1, 40, 40), (576, 1, 75, 17),
                                      (256, 3, 1, 5), (576, 1, 109, 216)])
def test_dwconv5x5_ring_is_bit_identical(lib, Cp, B, H, W):
    """dwconv_ring.hip (wave-private LDS-DMA ring, one wave per SIMD, no barriers) against dwconv5x5_kernel: same
    arithmetic in the same order per accumulator, so the outputs must agree bit for bit - image borders, widths that are
    not multiples of the 16-column wave tile, strips shorter than the ring depth, several strips per image, B > 1; run
    three times (the kernel's correctness rests on counted vmcnt waits: timing-dependent hazards)."""
    x, w, b = rnd(B, H, W, Cp, seed=1).bfloat16().cuda(), rnd(25, Cp, seed=2, std=0.2).cuda(), rnd(Cp, seed=3).cuda()
    outs = {}
    try:
        for mode in (0, 2, 2, 2):  # 2 = the ring kernel for every shape it takes (1, the default, keeps small launches on the old kernel)
            lib.roma_tuning(b"dw_ring", mode)
            out = torch.full((B, H, W, Cp), float("nan"), device="cuda", dtype=torch.bfloat16)
            ok(lib, lib.roma_op_dwconv5x5(P(x), P(out), P(w), P(b), B, H, W, Cp, BF16, None))
            torch.cuda.synchronize()
            if mode in outs:
                assert torch.equal(out.view(torch.int16), outs[mode].view(torch.int16))
            outs[mode] = out
    finally:
        lib.roma_tuning(b"dw_ring", -1)
    assert torch.isfinite(outs[2].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16)), float((outs[0].float() - outs[2].float()).abs().max())


def test_dwconv5x5_ring_repeated_launches_under_load(lib):
    """The ring kernel's only ordering between a wave's LDS-DMA and its LDS reads is a counted s_waitcnt: 200 launches of a
    benchmark-sized problem (16 x 108 x 108 x 1152: ~2 300 workgroups, 4.5 rounds), back to back with a GEMM on a second
    stream competing for the memory system, must all reproduce the register-prefetch kernel's output bit for bit."""
    B, H, W, Cp = 16, 108, 108, 1152
    x, w, b = rnd(B, H, W, Cp, seed=11).bfloat16().cuda(), rnd(25, Cp, seed=12, std=0.2).cuda(), rnd(Cp, seed=13).cuda()
    A, Wg = rnd(8192, 1024, seed=14).bfloat16().cuda(), rnd(1024, 1024, seed=15, std=0.03).bfloat16().cuda()
    Cg = torch.empty(8192, 1024, device="cuda", dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    try:
        lib.roma_tuning(b"dw_ring", 0)
        ref = torch.empty((B, H, W, Cp), device="cuda", dtype=torch.bfloat16)
        ok(lib, lib.roma_op_dwconv5x5(P(x), P(ref), P(w), P(b), B, H, W, Cp, BF16, None))
        torch.cuda.synchronize()
        lib.roma_tuning(b"dw_ring", 1)
        out = torch.empty_like(ref)
        bad = 0
        for it in range(200):
            if it % 4 == 0:
                ok(lib, lib.roma_op_gemm(P(A), 1024, P(Wg), 1024, P(Cg), 1024, 8192, 1024, 1024, 1, 0, 0, 0, None, None, None, 0, 0, 1.0,
                                         BF16, BF16, C.c_void_p(side.cuda_stream)))
            out.fill_(float("nan"))
            ok(lib, lib.roma_op_dwconv5x5(P(x), P(out), P(w), P(b), B, H, W, Cp, BF16, None))
            bad += int(not torch.equal(out.view(torch.int16), ref.view(torch.int16)))
        torch.cuda.synchronize()
    finally:
        lib.roma_tuning(b"dw_ring", -1)
    assert bad == 0, f"{bad} of 200 launches differ"


@pytest.mark.parametrize("Cp", [24, 144])
def test_refiner_block_repeated_launches_under_load(lib, Cp):
    """refiner_block24_wave_kernel / refiner_block144_1b_kernel order their rings with counted vmcnt waits (and one barrier
    per row); a wrong count shows up as a timing dependent mismatch, so: 150 launches while a GEMM runs on a second stream,
    every one bit-identical to the same kernel's result on an otherwise idle GPU (rounds 3-4 compared with the two-barrier
    workgroup kernel, which left the shipped library in round 5; test_refiner_block_fused holds the values to torch f64)."""
    B, H, W = (4, 211, 333) if Cp == 24 else (3, 150, 187)
    x = rnd(B, H, W, Cp, seed=1).to(torch.bfloat16).cuda()
    w, b = (rnd(25, Cp, seed=2, std=0.2)).cuda(), rnd(Cp, seed=3).cuda()
    pw, pb = rnd(Cp, Cp, seed=4, std=Cp ** -0.5).to(torch.bfloat16).cuda(), rnd(Cp, seed=5).cuda()
    ref = torch.empty_like(x)
    torch.cuda.synchronize()
    ok(lib, lib.roma_op_refiner_block(P(x), P(ref), P(w), P(b), P(pw), P(pb), B, H, W, Cp, BF16, None))
    torch.cuda.synchronize()
    A = rnd(8192, 1024, seed=6).to(torch.bfloat16).cuda()
    Wg = rnd(1024, 1024, seed=7, std=0.03).to(torch.bfloat16).cuda()
    Cg = torch.empty((8192, 1024), device="cuda", dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    out = torch.empty_like(ref)
    bad = 0
    for it in range(150):
        if it % 4 == 0:
            ok(lib, lib.roma_op_gemm(P(A), 1024, P(Wg), 1024, P(Cg), 1024, 8192, 1024, 1024, 1, 0, 0, 0, None, None, None, 0, 0, 1.0,
                                     BF16, BF16, C.c_void_p(side.cuda_stream)))
        out.fill_(float("nan"))
        ok(lib, lib.roma_op_refiner_block(P(x), P(out), P(w), P(b), P(pw), P(pb), B, H, W, Cp, BF16, None))
        bad += int(not torch.equal(out.view(torch.int16), ref.view(torch.int16)))
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of 150 launches differ"


@pytest.mark.parametrize("Cp,B,H,W", [(24, 2, 13, 10), (24, 1, 75, 301), (24, 1, 290, 150), (144, 2, 13, 10), (144, 1, 41, 59),
                                      (144, 1, 262, 31), (144, 5, 70, 280), (24, 3, 3, 200), (144, 2, 1, 30), (24, 2, 97, 40),
                                      (24, 1, 33, 81), (24, 1, 1, 21), (24, 4, 140, 139)])
def test_refiner_block_fused(lib, Cp, B, H, W):
    """Fused dw5x5+BN+ReLU+1x1 (refiner_block.hip) vs torch f64 on the same bf16-rounded operands: ragged strips,
    x tiles and pixel blocks, both strip heights (H >= 256 selects 36-row strips), strips shorter than the pipeline depth.
    Run three times: the results must agree bit for bit (timing-dependent races)."""
    x = rnd(B, Cp, H, W, seed=1).to(torch.bfloat16)
    w, b = rnd(Cp, 1, 5, 5, seed=2, std=0.2), rnd(Cp, seed=3)
    pw = rnd(Cp, Cp, seed=4, std=Cp ** -0.5).to(torch.bfloat16)
    pb = rnd(Cp, seed=5)
    mid = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=2, groups=Cp)).to(torch.bfloat16)
    ref = (F.conv2d(mid.double(), pw.double()[:, :, None, None], pb.double())).permute(0, 2, 3, 1)
    out = torch.full((B, H, W, Cp), float("nan"), device="cuda", dtype=torch.bfloat16)
    wp = w.reshape(Cp, 25).T.contiguous().cuda()
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    outs = []
    for _ in range(3):
        out = torch.full((B, H, W, Cp), float("nan"), device="cuda", dtype=torch.bfloat16)
        ok(lib, lib.roma_op_refiner_block(P(xin), P(out), P(wp), P(b.cuda()), P(pw.cuda()), P(pb.cuda()), B, H, W, Cp, BF16, None))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    out = outs[0]
    # the A/B variants left the shipped library in round 5 (make TOOLS=1 brings them back): asking for one fails loudly
    key = b"rb24w" if Cp == 24 else b"rb144_1b"
    lib.roma_tuning(key, 0)
    try:
        o2 = torch.empty((B, H, W, Cp), device="cuda", dtype=torch.bfloat16)
        rc = lib.roma_op_refiner_block(P(xin), P(o2), P(wp), P(b.cuda()), P(pw.cuda()), P(pb.cuda()), B, H, W, Cp, BF16, None)
        if rc == 0:  # a tools build: the two-barrier kernel ran and must give the same bits
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int16), o2.view(torch.int16))
        else:
            assert Cp == 144 and b"TOOLS" in lib.roma_last_error()
    finally:
        lib.roma_tuning(key, -1)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    # bf16 output rounding (2^-8 relative) + the occasional 1-ulp flip of the bf16 intermediate
    err = (got - ref).abs()
    assert (err <= 1e-2 * ref.abs() + 3e-2).all(), float(err.max())
    assert float(err.mean()) < 6e-3
    # unsupported configurations fail loudly instead of falling back
    assert lib.roma_op_refiner_block(P(out), P(out), P(wp), P(b.cuda()), P(pw.cuda()), P(pb.cuda()), B, H, W, Cp, BF16, None) != 0
    assert lib.roma_op_refiner_block(P(x.cuda()), P(out), P(wp), P(b.cuda()), P(pw.cuda()), P(pb.cuda()), B, H, W, Cp, F32, None) != 0


@pytest.mark.parametrize("dt,Cp,M", [(BF16, 24, 100003), (BF16, 24, 255), (BF16, 144, 30011), (BF16, 576, 5000), (F32, 24, 40001),
                                      (F32, 1152, 777)])
def test_refiner_out_conv_update(lib, dt, Cp, M):
    """out_conv (3 x Cp, f32) fused with the running flow / certainty update - the lane-per-row kernel (bf16, Cp = 24) and the
    shared-row kernels against f64."""
    import ctypes as C
    tdt = torch.float32 if dt == F32 else torch.bfloat16
    d = rnd(M, Cp, seed=1).to(tdt)
    w, b = rnd(3, Cp, seed=2, std=Cp ** -0.5), rnd(3, seed=3)
    flow0, cert0 = rnd(M, 2, seed=4), rnd(M, seed=5)
    sx, sy = 0.25, 0.125
    o = d.double() @ w.double().T + b.double()
    ref_flow = flow0.double() + o[:, :2] * torch.tensor([sx, sy], dtype=torch.float64)
    ref_cert = cert0.double() + o[:, 2]
    flow, cert = flow0.cuda(), cert0.cuda()
    ok(lib, lib.roma_op_refiner_out(P(d.cuda()), Cp, dt, P(w.cuda()), P(b.cuda()), P(flow), P(cert), M, Cp, C.c_float(sx), C.c_float(sy), None))
    torch.cuda.synchronize()
    tol = 2e-5 * max(1.0, (Cp / 24) ** 0.5)
    assert torch.allclose(flow.cpu().double(), ref_flow, atol=tol, rtol=1e-5)
    assert torch.allclose(cert.cpu().double(), ref_cert, atol=4 * tol, rtol=1e-5)


def test_maxpool_and_first_conv(lib):
    B, H, W = 2, 16, 24
    img, w, b = rnd(B, 3, H, W, seed=1), rnd(64, 3, 3, 3, seed=2, std=0.3), rnd(64, seed=3)
    ref = F.relu(F.conv2d(img.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    wp = w.permute(1, 2, 3, 0).reshape(27, 64).contiguous().cuda()
    out = torch.empty((B, H, W, 64), device="cuda")
    ok(lib, lib.roma_op_conv3x3_c3(P(img.cuda()), P(wp), P(b.cuda()), P(out), B, H, W, F32, None))
    pooled = torch.empty((B, H // 2, W // 2, 64), device="cuda")
    ok(lib, lib.roma_op_maxpool2x2(P(out), P(pooled), B, H, W, 64, F32, None))
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu().double(), ref, atol=1e-5, rtol=1e-5)
    refp = F.max_pool2d(ref.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.allclose(pooled.cpu().double(), refp, atol=1e-5)



@pytest.mark.parametrize("C,N,ldf", [(64, 9, 16), (128, 64, 64)])
@pytest.mark.parametrize("B,H,W", [(2, 16, 24), (1, 9, 33), (3, 37, 70), (2, 140, 216)])
def test_pool_proj_fused_is_maxpool_plus_proj_gemm(lib, C, N, ldf, B, H, W):
    """pool_proj.hip (round 6): MaxPool2d(2) and the proj head of a VGG level (Conv2d 1x1 + folded BN, roma_models.py:156-160) in
    one pass over the un-pooled map - bit-identical to the two kernels it replaces (roma_op_maxpool2x2, roma_op_gemm) and equal
    to torch on the same bf16 operands; odd H / W (floor pooling, ragged 32-pixel column tiles, a last row without a partner)."""
    x = F.relu(rnd(B, H, W, C, seed=1)).to(torch.bfloat16)                     # a post-ReLU map: never negative
    pw = rnd(N, C, seed=2, std=C ** -0.5).to(torch.bfloat16)
    pb = rnd(N, seed=3)
    xd, pwd, pbd = x.cuda(), pw.cuda(), pb.cuda()
    pooled = torch.full((B, H // 2, W // 2, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    pf = torch.full((B, H * W, ldf), float("nan"), device="cuda", dtype=torch.bfloat16)
    ok(lib, lib.roma_op_pool_proj(P(xd), P(pooled), P(pf), P(pwd), C, P(pbd), N, ldf, B, H, W, C, BF16, None))
    pooled2 = torch.empty_like(pooled)
    pf2 = torch.zeros_like(pf)
    ok(lib, lib.roma_op_maxpool2x2(P(xd), P(pooled2), B, H, W, C, BF16, None))
    ok(lib, lib.roma_op_gemm(P(xd), C, P(pwd), C, P(pf2), ldf, B * H * W, N, C, 1, 0, 0, 0, P(pbd), None, None, 0, 0, 1.0, BF16, BF16, None))
    torch.cuda.synchronize()
    assert torch.equal(pooled.view(torch.int16), pooled2.view(torch.int16))
    assert torch.equal(pf[:, :, :N].contiguous().view(torch.int16), pf2[:, :, :N].contiguous().view(torch.int16))
    assert float(pf[:, :, N:].float().abs().max()) == 0.0 if ldf > N else True   # the pad columns are written as zeros
    refp = F.max_pool2d(x.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(pooled.cpu().float(), refp)
    ref = x.double().reshape(B, H * W, C) @ pw.double().T + pb.double()
    assert torch.allclose(pf[:, :, :N].cpu().double(), ref, atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("B,H,W", [(2, 16, 24), (1, 9, 33), (3, 37, 70), (1, 560, 560), (2, 100, 864)])
def test_first_conv_bf16_fused(lib, B, H, W):
    """conv64.hip conv3x3_c3_bf16: the first VGG layer of the bf16 path straight from the f32 image (27 taps split over the two
    lanes of an MFMA column, K index permuted on both operands).  Against conv2d on the bf16-rounded operands in f64: the
    products are exact, the sum is f32, the output one bf16 rounding - and bit-for-bit against the im2col + GEMM pair it
    replaces is NOT required (the K order inside the MFMA differs), so that pair is held to the same reference instead.
    Image borders, ragged last column tile (W % 32 != 0), ragged last row block (H % 8 != 0)."""
    img, w, b = rnd(B, 3, H, W, seed=1), rnd(64, 3, 3, 3, seed=2, std=0.3), rnd(64, seed=3)
    imq, wq = img.bfloat16(), w.bfloat16()
    ref = F.relu(F.conv2d(imq.double(), wq.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    wp = torch.zeros(64, 32, dtype=torch.bfloat16)
    wp[:, :27] = wq.reshape(64, 27)  # k = ci*9 + ky*3 + kx
    out = torch.full((B, H, W, 64), -3.0, device="cuda", dtype=torch.bfloat16)
    for _ in range(2):
        ok(lib, lib.roma_op_conv3x3_c3_bf16(P(img.cuda()), P(wp.cuda()), P(b.cuda()), P(out), B, H, W, None))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs()
    assert float((err / (ref.abs() + 1.0)).max()) <= 2.0 ** -8 + 1e-6, float(err.max())


# ------------------------------------------------------------------ sampling: KDE + RegressionMatcher.sample
def test_kde_vs_reference_golden_and_oracle():
    import roma_amd
    from oracle import roma_oracle as O
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "kde_reference.npz"))
    x = torch.from_numpy(g["x"])
    xd = x.cuda()
    for key, kw in (("density_f32", {"half": False}), ("density_f32_down3", {"half": False, "down": 3}),
                    ("density_f32_std025", {"half": False, "std": 0.25})):
        d = roma_amd.kde(xd, **kw).cpu().numpy()
        r = g[key]                                   # the reference itself (f32 evaluation)
        assert np.all(np.abs(d - r) <= 1e-4 * np.abs(r) + 1e-6), key
        o = O.kde(x, **{k: v for k, v in kw.items() if k != "half"}).numpy()
        assert np.all(np.abs(d - o) <= 1e-5 * np.abs(o) + 1e-7), key  # f32 sum + v_exp_f32 vs the f64 oracle
    d = roma_amd.kde(xd).cpu().numpy()              # default half=True: same input rounding as the oracle
    o = O.kde(x, half=True).numpy()
    assert np.all(np.abs(d - o) <= 1e-5 * np.abs(o) + 1e-7)
    r = g["density_half"]
    assert np.all(np.abs(d - r) <= 0.15 * np.abs(r) + 0.15)  # the reference's own fp16 arithmetic noise


def test_kde_full_size_properties():
    """n = 40 000 (4 x num, the size matcher.sample evaluates): spot rows against float64, density >= 1, and
    invariance under a permutation of the points."""
    import roma_amd
    g = np.random.Generator(np.random.PCG64(5))
    n = 40000
    x = (g.random((n, 4), dtype=np.float32) * 2 - 1)
    x[: n // 2] = x[:50][g.integers(0, 50, n // 2)] + 0.03 * g.standard_normal((n // 2, 4), dtype=np.float32)
    xd = torch.from_numpy(x).cuda()
    d = roma_amd.kde(xd, half=False).cpu().numpy()
    assert d.shape == (n,) and np.all(d >= 1.0 - 1e-5)
    rows = g.integers(0, n, 64)
    ref = np.exp(-((x[rows, None, :].astype(np.float64) - x[None, :, :].astype(np.float64)) ** 2).sum(-1) / 0.02).sum(-1)
    assert np.all(np.abs(d[rows] - ref) <= 2e-5 * ref)
    perm = g.permutation(n)
    dp = roma_amd.kde(torch.from_numpy(x[perm]).cuda(), half=False).cpu().numpy()
    assert np.all(np.abs(dp - d[perm]) <= 1e-5 * d[perm])
    with pytest.raises(Exception):
        roma_amd.kde(torch.from_numpy(x))  # CPU tensor: no fallback


def test_sample_distribution_matches_oracle():
    """RegressionMatcher.sample on the device vs the oracle restatement: same control flow, distributional parity."""
    from roma_amd.matcher import RegressionMatcher
    from oracle import roma_oracle as O
    gen = torch.Generator().manual_seed(3)
    dense = torch.tensor([0.3, -0.2, 0.1, 0.4]) + 0.02 * torch.randn(3000, 4, generator=gen)
    loose = torch.tensor([-0.5, 0.5, -0.4, -0.3]) + 0.08 * torch.randn(1000, 4, generator=gen)
    matches = torch.cat([dense, loose])
    cert = torch.full((4000,), 0.5)
    cert[:10] = 0.01
    m = RegressionMatcher.__new__(RegressionMatcher)  # sample() needs only the two sampling attributes
    m.sample_mode, m.sample_thresh = "threshold_balanced", 0.05
    torch.manual_seed(11)
    fr = []
    for _ in range(4):
        gm, gc = m.sample(matches.cuda().reshape(40, 100, 4), cert.cuda().reshape(40, 100), num=500)
        assert gm.shape == (500, 4) and gc.shape == (500,) and gm.is_cuda
        fr.append(float((gm[:, 0] < -0.1).float().mean()))
    fo = []
    for _ in range(4):
        om, _ = O.sample(matches, cert, num=500, generator=gen)
        fo.append(float((om[:, 0] < -0.1).float().mean()))
    assert abs(np.mean(fr) - np.mean(fo)) < 0.08, (fr, fo)
    assert np.mean(fr) > 0.5
    m.sample_mode = "threshold"
    gm, gc = m.sample(matches.cuda(), cert.cuda(), num=500)
    assert gm.shape == (500, 4)
    vals = set(np.unique(gc.cpu().numpy()).tolist())
    assert vals <= {1.0, np.float32(0.01).item()}


def test_multinomial_without_replacement():
    """roma_amd.multinomial (exponential race + radix select) against the semantics of torch.multinomial(replacement=False):
    k distinct indices, never a zero-weight entry while positive ones remain, exactly the positive set when k equals their
    number, inclusion frequencies equal to torch's own (CPU) over many draws, reproducible from torch.manual_seed."""
    from roma_amd import multinomial
    g = torch.Generator().manual_seed(5)
    n = 20000
    w = torch.rand(n, generator=g) ** 3
    w[::7] = 0.0
    wd = w.cuda()
    torch.manual_seed(123)
    a = multinomial(wd, 3000)
    assert a.dtype == torch.int64 and a.shape == (3000,) and len(torch.unique(a)) == 3000
    assert bool((w[a.cpu()] > 0).all())
    torch.manual_seed(123)
    assert torch.equal(torch.sort(multinomial(wd, 3000)).values, torch.sort(a).values)
    npos = int((w > 0).sum())
    allpos = multinomial(wd, npos)
    assert set(allpos.cpu().tolist()) == set(torch.nonzero(w > 0)[:, 0].tolist())
    assert len(torch.unique(multinomial(wd, n))) == n  # k = n: a permutation (zero weights complete the sample)
    # inclusion frequencies on a small problem: ours vs torch CPU vs each other (sequential draws without replacement)
    ws = torch.tensor([8.0, 4.0, 2.0, 1.0, 1.0, 0.5, 0.25, 0.0])
    cnt_ours, cnt_ref, reps = torch.zeros(8), torch.zeros(8), 4000
    gen = torch.Generator().manual_seed(9)
    wsd = ws.cuda()
    for _ in range(reps):
        cnt_ours[multinomial(wsd, 3, generator=gen).cpu()] += 1
        cnt_ref[torch.multinomial(ws, 3, replacement=False, generator=gen)] += 1
    fo, fr = cnt_ours / reps, cnt_ref / reps
    assert fo[7] == 0 and float((fo - fr).abs().max()) < 0.03, (fo, fr)
    # big-n distribution: mean weight of the chosen set tracks torch's
    torch.manual_seed(7)
    mo = float(w[multinomial(wd, 2000).cpu()].mean())
    mr = float(w[torch.multinomial(w, 2000, replacement=False)].mean())
    assert abs(mo - mr) < 0.03 * mr + 0.01, (mo, mr)
    # draw order (torch.multinomial's): the FIRST draw follows the weights themselves, and a prefix is not index-sorted
    cnt_first, cnt_first_ref = torch.zeros(8), torch.zeros(8)
    for _ in range(reps):
        cnt_first[int(multinomial(wsd, 3, generator=gen)[0])] += 1
        cnt_first_ref[int(torch.multinomial(ws, 3, replacement=False, generator=gen)[0])] += 1
    assert float((cnt_first / reps - ws / ws.sum()).abs().max()) < 0.03, cnt_first / reps
    assert float((cnt_first - cnt_first_ref).abs().max()) / reps < 0.04
    big = multinomial(wd, 3000).cpu()
    assert not torch.equal(big, torch.sort(big).values)
    assert big[:300].float().std() > 0.2 * n  # the first 300 of 3000 are spread over the whole index range


def test_multinomial_large_k_keeps_the_draw_order():
    """k > 65 536 (sample(num) of 100 000 asks for 400 000): the draw order comes from the bitonic network instead of the all-pairs
    rank.  The race keys depend on (seed, index) only, so the first k' entries of a draw of k ARE the draw of k' with the same
    seed - checked against the all-pairs form at k' = 3 000 and 65 536, for a k that is and one that is not a power of two."""
    from roma_amd import multinomial
    g = torch.Generator().manual_seed(11)
    n = 600000
    w = torch.rand(n, generator=g) ** 2
    w[::5] = 0.0
    wd = w.cuda()
    for k in (70000, 131072, 400000):
        gen = torch.Generator().manual_seed(77)
        big = multinomial(wd, k, generator=gen)
        assert big.shape == (k,) and len(torch.unique(big)) == k and bool((w[big.cpu()] > 0).all())
        for kp in (3000, 65536):
            gen = torch.Generator().manual_seed(77)
            assert torch.equal(multinomial(wd, kp, generator=gen), big[:kp]), (k, kp)
    # more draws than positive weights: the zero-weight entries (key = +inf) complete the sample behind every positive one
    npos = int((w > 0).sum())
    gen = torch.Generator().manual_seed(3)
    allp = multinomial(wd, npos + 1000, generator=gen).cpu()
    assert len(torch.unique(allp)) == npos + 1000 and bool((w[allp[:npos]] > 0).all()) and bool((w[allp[npos:]] == 0).all())


def test_match_keypoints_vs_reference_golden():
    """RegressionMatcher.match_keypoints through roma_op_sample_warp_at + roma_op_mutual_nn: index-exact against the
    reference's own output (tests/golden/keypoints_reference.npz) for both parameter sets, plus the return variants."""
    from roma_amd.matcher import RegressionMatcher
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "keypoints_reference.npz"))
    t = {k: torch.from_numpy(g[k]).cuda() for k in ("warp", "cert", "x_A", "x_B")}
    m = RegressionMatcher.__new__(RegressionMatcher)
    for name, kw in (("default", {}), ("loose", dict(max_dist=0.02, cert_th=0.6))):
        iA, iB = m.match_keypoints(t["x_A"], t["x_B"], t["warp"], t["cert"], return_inds=True, **kw)
        assert np.array_equal(iA.cpu().numpy(), g["inds_A_" + name]), name
        assert np.array_equal(iB.cpu().numpy(), g["inds_B_" + name]), name
    kA, kB = m.match_keypoints(t["x_A"], t["x_B"], t["warp"], t["cert"])
    assert torch.equal(kA, t["x_A"][iA0 := torch.from_numpy(g["inds_A_default"]).cuda()]) and kB.shape == (len(iA0), 2)
    cat = m.match_keypoints(t["x_A"], t["x_B"], t["warp"], t["cert"], return_tuple=False)
    assert cat.shape == (len(iA0), 4)
    with pytest.raises(Exception):
        m.match_keypoints(t["x_A"].cpu(), t["x_B"], t["warp"], t["cert"])
    # empty keypoint sets do not launch anything
    e = m.match_keypoints(t["x_A"][:0], t["x_B"], t["warp"], t["cert"], return_inds=True)
    assert len(e[0]) == 0 and len(e[1]) == 0


def test_match_keypoints_ties_vs_reference_golden():
    """Duplicate keypoints: the reference's torch.nonzero returns every tied mutual pair in row-major order
    (matcher.py:756-762); roma_op_mutual_nn_count / _fill must return exactly that list."""
    from roma_amd.matcher import RegressionMatcher
    g = np.load(os.path.join(GOLDEN, "keypoints_ties_reference.npz"))
    t = {k: torch.from_numpy(g[k]).cuda() for k in ("warp", "cert", "x_A", "x_B")}
    m = RegressionMatcher.__new__(RegressionMatcher)
    for name, kw in (("default", {}), ("loose", dict(max_dist=0.02, cert_th=0.6))):
        iA, iB = m.match_keypoints(t["x_A"], t["x_B"], t["warp"], t["cert"], return_inds=True, **kw)
        assert len(g["inds_A_" + name]) > len(np.unique(g["inds_A_" + name]))  # the fixture does contain tied pairs
        assert np.array_equal(iA.cpu().numpy(), g["inds_A_" + name]), name
        assert np.array_equal(iB.cpu().numpy(), g["inds_B_" + name]), name


def test_visualize_warp_vs_reference_golden(tmp_path):
    """RegressionMatcher.visualize_warp on the device (roma_op_visualize_warp: grid_sample + certainty blend in one kernel)
    against the reference's own output (tests/golden/visualize_reference.npz): symmetric with tensor images, one
    direction with an image of another resolution; plus the save_path route."""
    from roma_amd import RegressionMatcher
    g = np.load(os.path.join(GOLDEN, "visualize_reference.npz"))
    t = {k: torch.from_numpy(g[k]).cuda() for k in g.files}
    m = RegressionMatcher.__new__(RegressionMatcher)  # the method needs no model handle
    W = t["im_A"].shape[-1]
    vs = m.visualize_warp(t["warp"], t["cert"], im_A=t["im_A"], im_B=t["im_B"], symmetric=True)
    assert vs.shape == t["vis_sym"].shape and float((vs - t["vis_sym"]).abs().max()) < 2e-6
    p = str(tmp_path / "vis.png")
    vo = m.visualize_warp(t["warp"][:, :W].contiguous(), t["cert"][:, :W].contiguous(), im_A=t["im_A"], im_B=t["im_B2"],
                          symmetric=False, save_path=p)
    assert float((vo - t["vis_one"]).abs().max()) < 2e-6 and os.path.getsize(p) > 0
    with pytest.raises(Exception):
        m.visualize_warp(t["warp"].cpu(), t["cert"].cpu(), im_A=t["im_A"], im_B=t["im_B"])


def test_fb_consistency_vs_oracle():
    """conf_from_fb_consistency on the device vs the oracle restatement (matcher.py:672-699).  The output is a hard
    threshold: pixels whose round-trip error is within 1e-5 of the threshold are excluded from the comparison."""
    from roma_amd.matcher import RegressionMatcher
    from oracle import roma_oracle as O
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 48, 64
    ys, xs = torch.meshgrid(torch.linspace(-1 + 1 / H, 1 - 1 / H, H), torch.linspace(-1 + 1 / W, 1 - 1 / W, W), indexing="ij")
    grid = torch.stack([xs, ys], dim=-1)[None].repeat(B, 1, 1, 1)
    fwd = grid + 0.05 * torch.sin(3 * grid.flip(-1)) + 0.02 * torch.randn(B, H, W, 2, generator=g)
    bwd = grid - 0.05 * torch.sin(3 * grid.flip(-1)) + 0.02 * torch.randn(B, H, W, 2, generator=g)
    ref = O.conf_from_fb_consistency(fwd, bwd, th=2)
    coords_fb = F.grid_sample(bwd.permute(0, 3, 1, 2), fwd, align_corners=False).permute(0, 2, 3, 1)
    margin = ((grid - coords_fb).norm(dim=-1) - 2 * 2 / max(H, W)).abs()
    m = RegressionMatcher.__new__(RegressionMatcher)
    got = m.conf_from_fb_consistency(fwd.cuda(), bwd.cuda(), th=2).cpu()
    assert got.shape == (B, H, W) and 0.2 < float(ref.mean()) < 0.98
    assert torch.equal(got[margin > 1e-5], ref[margin > 1e-5])
    one = m.conf_from_fb_consistency(fwd[0].cuda(), bwd[0].cuda(), th=2).cpu()
    assert one.shape == (H, W) and torch.equal(one, got[0])


@pytest.mark.parametrize("B,H,W", [(2, 13, 10), (1, 8, 16), (1, 41, 59), (3, 24, 33), (1, 140, 140), (2, 1, 30), (1, 75, 3)])
def test_refiner_block_wide_fused(lib, B, H, W):
    """refiner_block_wide.hip - the C = 576 ConvRefiner block (dw5x5 + BN + ReLU + 1x1, matcher.py:92-122) in ONE kernel, all
    output channels per workgroup - against torch f64 on the same bf16-rounded operands (ragged tiles, images smaller than a
    tile, every border), against the two-kernel path it replaces (dwconv5x5 + 1x1 GEMM: the depthwise half is bit-identical by
    construction, the 1x1 accumulates in f32 on another MFMA shape, so results may differ by an ulp of the 16-bit output), and
    against itself (three launches: timing-dependent hazards of its DMA / barrier schedule)."""
    Cp = 576
    x = rnd(B, Cp, H, W, seed=1).to(torch.bfloat16)
    w, b = rnd(Cp, 1, 5, 5, seed=2, std=0.2), rnd(Cp, seed=3)
    pw = rnd(Cp, Cp, seed=4, std=Cp ** -0.5).to(torch.bfloat16)
    pb = rnd(Cp, seed=5)
    mid = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=2, groups=Cp)).to(torch.bfloat16)
    ref = (F.conv2d(mid.double(), pw.double()[:, :, None, None], pb.double())).permute(0, 2, 3, 1)
    wp = w.reshape(Cp, 25).T.contiguous().cuda()
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    bd, pwd, pbd = b.cuda(), pw.cuda(), pb.cuda()
    outs = []
    for _ in range(3):
        out = torch.full((B, H, W, Cp), float("nan"), device="cuda", dtype=torch.bfloat16)
        ok(lib, lib.roma_op_refiner_block(P(xin), P(out), P(wp), P(bd), P(pwd), P(pbd), B, H, W, Cp, BF16, None))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)) and torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))
    got = outs[0].cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert (err <= 1e-2 * ref.abs() + 3e-2).all(), float(err.max())
    assert float(err.mean()) < 6e-3
    # the pair of kernels it replaces
    t = torch.empty_like(outs[0])
    y2 = torch.empty_like(outs[0])
    ok(lib, lib.roma_op_dwconv5x5(P(xin), P(t), P(wp), P(bd), B, H, W, Cp, BF16, None))
    ok(lib, lib.roma_op_gemm(P(t), Cp, P(pwd), Cp, P(y2), Cp, B * H * W, Cp, Cp, 1, 0, 0, 0, P(pbd), None, None, 0, 0, 1.0, BF16, BF16, None))
    torch.cuda.synchronize()
    # the stand-alone stencil accumulates in f32, `mid` is the f64 result rounded once: equal up to one bf16 ulp, and
    # almost everywhere bit for bit
    tm, mm = t.cpu().float(), mid.permute(0, 2, 3, 1).contiguous().float()
    assert float((tm - mm).abs().max()) <= 2.0 ** -7 * float(mm.abs().max()) + 1e-6
    assert float((tm.view(torch.int32) == mm.view(torch.int32)).float().mean()) > 0.98
    d = (outs[0].float() - y2.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(y2.float().abs().max()) + 1e-6, float(d.max())  # within one bf16 ulp of each other
    assert float((outs[0].view(torch.int16) == y2.view(torch.int16)).float().mean()) > 0.98


def test_refiner_block_wide_repeated_launches_under_load(lib):
    """150 launches of the fused C = 576 block while a GEMM runs on a second stream: every one bit-identical to the quiet
    launch (the kernel orders its LDS-DMA with vmcnt(0) + workgroup barriers; a hazard would be timing dependent)."""
    B, H, W, Cp = 4, 140, 140, 576
    x = rnd(B, H, W, Cp, seed=1).to(torch.bfloat16).cuda()
    w, b = (rnd(25, Cp, seed=2, std=0.2)).cuda(), rnd(Cp, seed=3).cuda()
    pw, pb = rnd(Cp, Cp, seed=4, std=Cp ** -0.5).to(torch.bfloat16).cuda(), rnd(Cp, seed=5).cuda()
    ref = torch.empty_like(x)
    torch.cuda.synchronize()
    ok(lib, lib.roma_op_refiner_block(P(x), P(ref), P(w), P(b), P(pw), P(pb), B, H, W, Cp, BF16, None))
    torch.cuda.synchronize()
    A = rnd(8192, 1024, seed=6).to(torch.bfloat16).cuda()
    Wg = rnd(1024, 1024, seed=7, std=0.03).to(torch.bfloat16).cuda()
    Cg = torch.empty((8192, 1024), device="cuda", dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    out = torch.empty_like(ref)
    bad = 0
    for it in range(150):
        if it % 4 == 0:
            ok(lib, lib.roma_op_gemm(P(A), 1024, P(Wg), 1024, P(Cg), 1024, 8192, 1024, 1024, 1, 0, 0, 0, None, None, None, 0, 0, 1.0,
                                     BF16, BF16, C.c_void_p(side.cuda_stream)))
        out.fill_(float("nan"))
        ok(lib, lib.roma_op_refiner_block(P(x), P(out), P(w), P(b), P(pw), P(pb), B, H, W, Cp, BF16, None))
        bad += int(not torch.equal(out.view(torch.int16), ref.view(torch.int16)))
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of 150 launches differ"


@pytest.mark.parametrize("Cp,B,H,W", [(24, 2, 13, 10), (24, 1, 75, 301), (24, 3, 3, 200), (24, 1, 290, 150), (144, 2, 13, 10),
                                      (144, 1, 41, 59), (144, 1, 262, 31), (144, 2, 1, 30), (144, 3, 70, 140)])
def test_refiner_block_final_composed_out_conv(lib, Cp, B, H, W):
    """The FINAL form of the fused narrow ConvRefiner blocks (round 5): depthwise 5x5 + BN + ReLU, then the block's 1x1 composed
    with out_conv (linear o linear, matcher.py:92-122, 175-178) as ONE C -> 3 map on the MFMA - weights as a 16-bit head + 16-bit
    remainder in rows 0-2 / 4-6 of an 8-row matrix, bias in the accumulator - written as per-pixel deltas, and the pass that adds
    them to flow / certainty.  Against torch f64 on the same 16-bit-rounded depthwise output; three launches agree bit for bit."""
    x = rnd(B, Cp, H, W, seed=1).to(torch.bfloat16)
    w, b = rnd(Cp, 1, 5, 5, seed=2, std=0.2), rnd(Cp, seed=3)
    wc = rnd(3, Cp, seed=4, std=Cp ** -0.5)          # the composed out_w . pw8 (f32 at pack time)
    bc = rnd(3, seed=5)
    hi = wc.to(torch.bfloat16)
    lo = (wc - hi.float()).to(torch.bfloat16)
    pwf = torch.zeros(8, Cp, dtype=torch.bfloat16)
    pwf[0:3], pwf[4:7] = hi, lo
    bf = torch.zeros(Cp)
    bf[:3] = bc
    wp = w.reshape(Cp, 25).T.contiguous().cuda()
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    # the depthwise half from the stand-alone kernel (same FMA order: its 16-bit output is what the fused kernel puts into LDS;
    # a torch f64 stencil rounds 1 in ~200 values to the neighbouring 16-bit number), checked against torch loosely
    t = torch.empty_like(xin)
    ok(lib, lib.roma_op_dwconv5x5(P(xin), P(t), P(wp), P(b.cuda()), B, H, W, Cp, BF16, None))
    torch.cuda.synchronize()
    mid = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=2, groups=Cp)).permute(0, 2, 3, 1)
    assert float((t.cpu().double() - mid).abs().max()) <= 2.0 ** -7 * float(mid.abs().max())
    ref = torch.einsum("bhwc,oc->bhwo", t.cpu().double(), hi.double() + lo.double()) + bc.double()
    outs = []
    for _ in range(3):
        delta = torch.full((B, H, W, 4), float("nan"), device="cuda")
        ok(lib, lib.roma_op_refiner_block_final(P(xin), P(delta), P(wp), P(b.cuda()), P(pwf.cuda()), P(bf.cuda()), B, H, W, Cp, BF16, None))
        torch.cuda.synchronize()
        outs.append(delta)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    got = outs[0].cpu().double()
    assert torch.isfinite(got).all() and float(got[..., 3].abs().max()) == 0.0
    err = (got[..., :3] - ref).abs()
    scale = float(ref.abs().max())
    print(f"refiner_block_final C={Cp}: max |delta - f64| = {float(err.max()):.2e} (|delta| max {scale:.2f})")
    assert float(err.max()) < 2e-5 * max(scale, 1.0) + 1e-5, (float(err.max()), scale)
    # the apply pass
    flow, cert = rnd(B * H * W, 2, seed=6).cuda(), rnd(B * H * W, seed=7).cuda()
    f0, c0 = flow.clone(), cert.clone()
    ok(lib, lib.roma_op_refiner_apply_delta(P(outs[0]), P(flow), P(cert), B * H * W, 0.25, 0.5, None))
    torch.cuda.synchronize()
    d = outs[0].reshape(-1, 4)
    assert torch.equal(flow[:, 0], f0[:, 0] + 0.25 * d[:, 0]) and torch.equal(flow[:, 1], f0[:, 1] + 0.5 * d[:, 1])
    assert torch.equal(cert, c0 + d[:, 2])
