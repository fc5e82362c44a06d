"""libroma_hip_f16.so - the same sources built for IEEE binary16 storage (amp_dtype=torch.float16, the reference's default
precision policy: model_zoo/__init__.py:37, matcher.py:46,341, encoders.py:7) - operator parity on MI355X.

Every format-specific piece of the library (conversions, packing, v_mfma_f32_32x32x16_f16, v_dot2_f32_f16) sits in
csrc/common.h; these tests run one operator of every kernel family through the f16 library against torch f64 on the same
f16-rounded operands, with bounds 8 x tighter than the bf16 ones (11 significand bits instead of 8).  The model-level gates
are tests/test_gpu_parity.py::test_h16_tiny_stagewise_vs_oracle[f16] and ::test_f16_full8_vs_reference_golden."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import P, ok, rnd

pytestmark = pytest.mark.gpu

F32, BF16, F16 = 0, 1, 2


@pytest.fixture(scope="module")
def lib16(built_lib):
    from roma_amd import _lib
    lib = _lib.load("f16")
    assert lib.roma_h16_format() == F16 and b"f16" in lib.roma_version()
    return lib


def test_each_library_rejects_the_other_16_bit_code(built_lib, lib16):
    """The 16-bit format is a build property: the other code is an error, never a reinterpretation of the bits."""
    a = torch.zeros(64, 64, device="cuda", dtype=torch.float16)
    o = torch.zeros(64, 64, device="cuda", dtype=torch.float16)
    for lib, bad in ((lib16, BF16), (built_lib, F16)):
        rc = lib.roma_op_gemm(P(a), 64, P(a), 64, P(o), 64, 64, 64, 64, 1, 0, 0, 0, None, None, None, 0, 0, 1.0, bad, bad, None)
        assert rc != 0 and b"this library stores" in lib.roma_last_error()
    from roma_amd import _lib
    cfg = _lib.RomaConfig(112, 112, 0, 0, 1, 0, 1, BF16, 1, 0)
    h = C.c_void_p()
    assert lib16.roma_create(C.byref(cfg), C.byref(h)) != 0


@pytest.mark.parametrize("M,N,K,act", [(300, 200, 72, 0), (2500, 144, 144, 1), (9000, 768, 1024, 2), (25616, 1024, 1024, 0),
                                       (70000, 576, 576, 1), (8300, 1152, 1152, 0), (512, 1024, 4096, 0)])
def test_gemm_f16(lib16, M, N, K, act):
    """classic / 8-phase / 6-phase kernels: f16 in, f32 and f16 out, bias + activation"""
    A, W, b = rnd(M, K, seed=1).half(), rnd(N, K, seed=2, std=K ** -0.5).half(), rnd(N, seed=3)
    ref = A.double() @ W.double().T + b.double()
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    for dt_out, tdt, tol in ((F32, torch.float32, 2e-4), (F16, torch.float16, 2.5e-3)):
        out = torch.empty((M, N), device="cuda", dtype=tdt)
        ok(lib16, lib16.roma_op_gemm(P(Ad), K, P(Wd), K, P(out), N, M, N, K, 1, 0, 0, 0, P(bd), None, None, 0, act, 1.0, F16, dt_out, None))
        torch.cuda.synchronize()
        err = (out.cpu().double() - ref).abs()
        assert bool((err <= tol * (1.0 + ref.abs())).all()), (dt_out, float(err.max()))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 70, 70, 256, 512), (2, 72, 70, 64, 64), (2, 36, 40, 64, 128), (1, 54, 54, 128, 128)])
def test_conv3x3_f16(lib16, B, H, W, Cin, Cout):
    """implicit-GEMM 3x3 (gemm8p) and the weight-stationary front-end kernels (conv64.hip) in f16"""
    x, w, b = rnd(B, Cin, H, W, seed=1).half(), rnd(Cout, Cin, 3, 3, seed=2, std=(9 * Cin) ** -0.5).half(), rnd(Cout, seed=3)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()
    out = torch.zeros((B, H, W, Cout), device="cuda", dtype=torch.float16)
    ok(lib16, lib16.roma_op_conv3x3(P(xin), P(wp), P(b.cuda()), P(out), B, H, W, Cin, Cout, 1, F16, None))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs()
    assert bool((err <= 2.5e-3 * (1.0 + ref.abs())).all()), float(err.max())


def test_conv1_1_from_f32_image_f16(lib16):
    B, H, W = 2, 40, 56
    x, w, b = rnd(B, 3, H, W, seed=1), rnd(64, 3, 3, 3, seed=2, std=0.2).half(), rnd(64, seed=3)
    ref = F.relu(F.conv2d(x.half().double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    wp = torch.zeros(64, 32, dtype=torch.float16)
    wp[:, :27] = w.reshape(64, 27)
    out = torch.zeros((B, H, W, 64), device="cuda", dtype=torch.float16)
    ok(lib16, lib16.roma_op_conv3x3_c3_bf16(P(x.cuda()), P(wp.cuda()), P(b.cuda()), P(out), B, H, W, None))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs()
    assert bool((err <= 2.5e-3 * (1.0 + ref.abs())).all()), float(err.max())


@pytest.mark.parametrize("Cp,B,H,W", [(576, 2, 40, 36), (1152, 1, 27, 30)])
def test_dwconv5x5_f16(lib16, Cp, B, H, W):
    x, w, b = rnd(B, Cp, H, W, seed=1).half(), rnd(Cp, 1, 5, 5, seed=2, std=0.2), rnd(Cp, seed=3)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=2, groups=Cp)).permute(0, 2, 3, 1)
    wp = w.reshape(Cp, 25).T.contiguous().cuda()
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    outs = []
    try:
        for mode in (0, 2):  # register-prefetch kernel / wave-private ring kernel (dwconv_ring.hip): bit-identical
            lib16.roma_tuning(b"dw_ring", mode)
            out = torch.full((B, H, W, Cp), float("nan"), device="cuda", dtype=torch.float16)
            ok(lib16, lib16.roma_op_dwconv5x5(P(xin), P(out), P(wp), P(b.cuda()), B, H, W, Cp, F16, None))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib16.roma_tuning(b"dw_ring", -1)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    err = (outs[1].cpu().double() - ref).abs()
    assert bool((err <= 2.5e-3 * (1.0 + ref.abs())).all()), float(err.max())


@pytest.mark.parametrize("Cp,B,H,W", [(24, 1, 75, 301), (144, 1, 41, 59), (144, 2, 70, 280)])
def test_refiner_block_f16(lib16, Cp, B, H, W):
    x = rnd(B, Cp, H, W, seed=1).half()
    w, b = rnd(Cp, 1, 5, 5, seed=2, std=0.2), rnd(Cp, seed=3)
    pw, pb = rnd(Cp, Cp, seed=4, std=Cp ** -0.5).half(), rnd(Cp, seed=5)
    mid = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=2, groups=Cp)).half()
    ref = (F.conv2d(mid.double(), pw.double()[:, :, None, None], pb.double())).permute(0, 2, 3, 1)
    out = torch.full((B, H, W, Cp), float("nan"), device="cuda", dtype=torch.float16)
    wp = w.reshape(Cp, 25).T.contiguous().cuda()
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    ok(lib16, lib16.roma_op_refiner_block(P(xin), P(out), P(wp), P(b.cuda()), P(pw.cuda()), P(pb.cuda()), B, H, W, Cp, F16, None))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs()
    # f16 output rounding (2^-11 relative) + the occasional 1-ulp flip of the f16 intermediate
    assert bool((err <= 2e-3 * ref.abs() + 5e-3).all()), float(err.max())
    assert float(err.mean()) < 1e-3


@pytest.mark.parametrize("r,c,h,w", [(2, 256, 37, 43), (3, 512, 35, 35), (7, 512, 20, 24)])
def test_local_corr_window_f16(lib16, r, c, h, w):
    """tile + work-list kernels (v_dot2_f32_f16) through the Python operator boundary (float16 tensors select the f16
    library) against the oracle on f16-rounded features: smooth field, an incoherent band, far-outside targets."""
    from oracle import roma_oracle
    from roma_amd.local_correlation import local_correlation
    B = 2
    f0, f1 = rnd(B, c, h, w, seed=1), rnd(B, c, h, w, seed=2)
    warp = roma_oracle.pixel_grid(B, h, w) * 1.15 + 0.08 + rnd(B, 2, h, w, seed=3, std=0.004)
    warp[:, :, h // 2:h // 2 + 5] += rnd(B, 2, 5, w, seed=4, std=0.5)
    warp[1, :, :3, :3] = 3.0
    ref = roma_oracle.local_correlation(f0.half().float(), f1.half().float(), r, warp)
    out = local_correlation(f0.cuda().half(), f1.cuda().half(), r, warp.cuda())
    assert torch.allclose(out.cpu(), ref, atol=2e-4, rtol=1e-4), float((out.cpu() - ref).abs().max())
