"""End-to-end parity of the HIP match() path on a real MI355X, through the reference-shaped API
(roma_amd.roma_model(...).match(...)) which calls the C ABI:
  * against goldens produced by the UNMODIFIED reference on CPU (tests/golden/match_*.npz), and
  * against the CPU oracle (oracle/roma_oracle.py) run live, stage by stage.
Tolerance (BASELINE.json north_star): 1e-3 max-abs on (warp, certainty) in fp32 mode."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _to_dev(d):
    return {k: v.cuda() for k, v in d.items()}


def _nchw(a, b, h, w, c):
    """debug stage [b, h*w, c] channels-last (numpy) -> torch [b,c,h,w]"""
    return torch.from_numpy(a.reshape(b, h, w, c)).permute(0, 3, 1, 2)


@pytest.fixture(scope="module")
def tiny_model(built_lib, weights0):
    from roma_amd import roma_model
    sd, dsd = weights0
    m = roma_model((112, 112), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=torch.float32,
                   symmetric=True, upsample_res=(168, 168), max_batch=2)
    return m


def test_tiny_match_vs_reference_golden(tiny_model):
    from roma_amd import synthetic
    g = np.load(os.path.join(GOLDEN, "match_tiny.npz"))
    inp = _to_dev(synthetic.make_inputs(1, 112, 168, seed=1))
    warp, cert = tiny_model.match(inp["im_A"], inp["im_B"], im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
    torch.cuda.synchronize()
    assert warp.shape == (1, 168, 336, 4) and cert.shape == (1, 168, 336)
    dw = np.abs(warp.cpu().numpy() - g["warp"]).max()
    dc = np.abs(cert.cpu().numpy() - g["certainty"]).max()
    print(f"tiny: max|dwarp|={dw:.3e} max|dcert|={dc:.3e}")
    assert dw < TOL and dc < TOL


def test_resolutions_not_multiples_of_8_vs_reference_golden(built_lib, weights0):
    """roma_models.py:58-59 only asks for multiples of 14 (518, 574, 602, ...): coarse 126 x 154 -> upsample 182 x 198, where
    every level of the VGG pyramid has the floor-divided size of its max-pool (63 x 77, 31 x 38, 15 x 19 / 91 x 99, 45 x 49,
    22 x 24) and the decoder resizes go to those sizes.  f32 against the unmodified reference (tests/golden/match_odd.npz,
    tools/make_goldens.py odd) at 1e-3; the 16-bit modes must run and stay finite; the next multiple of 14 above 560 that
    is not a multiple of 8 (574) builds and runs."""
    from roma_amd import roma_model, synthetic
    sd, dsd = weights0
    g = np.load(os.path.join(GOLDEN, "match_odd.npz"))
    inp = _to_dev(synthetic.make_inputs(1, (126, 154), (182, 198), seed=5))
    kw = dict(im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
    m = roma_model((126, 154), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=torch.float32,
                   symmetric=True, upsample_res=(182, 198), max_batch=1)
    warp, cert = m.match(inp["im_A"], inp["im_B"], **kw)
    torch.cuda.synchronize()
    assert warp.shape == (1, 182, 396, 4) and cert.shape == (1, 182, 396)
    dw = np.abs(warp.cpu().numpy() - g["warp"]).max()
    dc = np.abs(cert.cpu().numpy() - g["certainty"]).max()
    print(f"odd: max|dwarp|={dw:.3e} max|dcert|={dc:.3e}")
    assert dw < TOL and dc < TOL
    for amp in (torch.bfloat16, torch.float16):
        mh = roma_model((126, 154), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=amp,
                        symmetric=True, upsample_res=(182, 198), max_batch=1)
        w16, c16 = mh.match(inp["im_A"], inp["im_B"], **kw)
        assert torch.isfinite(w16).all() and torch.isfinite(c16).all()
        assert float((c16 - cert).abs().mean()) < 0.05
    big = roma_model((574, 574), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=torch.bfloat16,
                     symmetric=True, upsample_res=(602, 602), max_batch=1)
    ib = _to_dev(synthetic.make_inputs(1, 574, 602, seed=2))
    wb, cb = big.match(ib["im_A"], ib["im_B"], im_A_high_res=ib["im_A_high_res"], im_B_high_res=ib["im_B_high_res"])
    assert wb.shape == (1, 602, 1204, 4) and torch.isfinite(wb).all() and torch.isfinite(cb).all()


def test_tiny_stages_vs_oracle(tiny_model, weights0):
    """Stage-by-stage comparison (debug capture) so that a regression names the kernel that broke."""
    from oracle import roma_oracle as O
    from roma_amd import synthetic
    sd, dsd = weights0
    inp = synthetic.make_inputs(1, 112, 168, seed=1)
    st = {}
    O.match(inp["im_A"], inp["im_B"], sd, dsd, inp["im_A_high_res"], inp["im_B_high_res"], stages=st)
    tiny_model.debug = True
    d = _to_dev(inp)
    try:
        tiny_model.match(d["im_A"], d["im_B"], im_A_high_res=d["im_A_high_res"], im_B_high_res=d["im_B_high_res"])
        torch.cuda.synchronize()
        report = {}

        def cmp(name, got, ref, tol):
            err = float((got - ref).abs().max())
            report[name] = err
            assert err < tol, f"stage {name}: max abs err {err:.3e} (tol {tol})"

        for s, c in ((1, 64), (2, 128), (4, 256), (8, 512)):
            h = 112 // s
            cmp(f"feat{s}", _nchw(tiny_model.debug_fetch(f"feat{s}"), 2, h, h, c), st[f"feat{s}"], 2e-4)
        cmp("feat16", _nchw(tiny_model.debug_fetch("feat16"), 2, 8, 8, 1024), st["feat16"], 5e-4)
        tok = tiny_model.debug_fetch("tokens16").reshape(2, 64, 1024)
        cmp("gp16", torch.from_numpy(tok[:, :, :512]).permute(0, 2, 1).reshape(2, 512, 8, 8), st["gp16"], 2e-4)
        logits = tiny_model.debug_fetch("logits16").reshape(2, 64, 4104)
        cls = torch.from_numpy(logits[:, :, :4096]).permute(0, 2, 1).reshape(2, 4096, 8, 8)
        cmp("cls16", cls, st["cls16"], 2e-3)
        assert torch.equal(cls.argmax(1), st["cls16"].argmax(1)), "coarse class arg-max differs from the oracle"
        cmp("gm_flow16", torch.from_numpy(tiny_model.debug_fetch("gm_flow16").reshape(2, 8, 8, 2)).permute(0, 3, 1, 2), st["gm_flow16"], 1e-4)
        for p, res in (("p1", 112), ("p2", 168)):
            for s in (16, 8, 4, 2, 1):
                if p == "p2" and s == 16:
                    continue
                h = 8 if s == 16 else res // s
                cmp(f"{p}_flow{s}", torch.from_numpy(tiny_model.debug_fetch(f"{p}_flow{s}").reshape(2, h, h, 2)).permute(0, 3, 1, 2),
                    st[f"{p}_flow{s}"], 2e-4)
                cmp(f"{p}_cert{s}", torch.from_numpy(tiny_model.debug_fetch(f"{p}_cert{s}").reshape(2, 1, h, h)), st[f"{p}_cert{s}"], 1e-3)
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(report, open("gpurun_out/stage_errors_tiny.json", "w"), indent=1)
    finally:
        tiny_model.debug = False


def test_mutable_attributes_and_batching(tiny_model, weights0):
    """symmetric / upsample_preds toggles (README.md:82-90, tests/test_match_modes.py:49-51) and B > max_batch chunking."""
    from oracle import roma_oracle as O
    from roma_amd import synthetic
    sd, dsd = weights0
    inp = synthetic.make_inputs(3, 112, 168, seed=11)
    d = _to_dev(inp)
    try:
        tiny_model.symmetric = False
        warp, cert = tiny_model.match(d["im_A"], d["im_B"], im_A_high_res=d["im_A_high_res"], im_B_high_res=d["im_B_high_res"])
        assert warp.shape == (3, 168, 168, 4) and cert.shape == (3, 168, 168)
        w_ref, c_ref = O.match(inp["im_A"], inp["im_B"], sd, dsd, inp["im_A_high_res"], inp["im_B_high_res"], symmetric=False)
        assert (warp.cpu() - w_ref).abs().max() < TOL and (cert.cpu() - c_ref).abs().max() < TOL
        tiny_model.symmetric = True
        tiny_model.upsample_preds = False
        warp, cert = tiny_model.match(d["im_A"], d["im_B"])
        assert warp.shape == (3, 112, 224, 4)
        w_ref, c_ref = O.match(inp["im_A"], inp["im_B"], sd, dsd, symmetric=True, upsample_preds=False)
        assert (warp.cpu() - w_ref).abs().max() < TOL and (cert.cpu() - c_ref).abs().max() < TOL
    finally:
        tiny_model.symmetric = True
        tiny_model.upsample_preds = True


def test_error_behaviour_matches_reference(tiny_model):
    x = torch.randn(1, 3, 112, 112, device="cuda")
    with pytest.raises(ValueError):  # matcher.py:791-792
        tiny_model.match(x, x, batched=False)
    with pytest.raises(AssertionError):  # matcher.py:543-545
        tiny_model.match(torch.randn(1, 3, 100, 112, device="cuda"), x)
    with pytest.raises(AssertionError):  # matcher.py:864: tensors + upsample_preds without high-res inputs
        tiny_model.match(x, x)
    with pytest.raises(ValueError):  # matcher.py:873-874: only one high-res image
        tiny_model.match(x, x, im_A_high_res=torch.randn(1, 3, 168, 168, device="cuda"))
    with pytest.raises(Exception):  # no CPU fallback
        tiny_model.match(x.cpu(), x.cpu())


def test_strict_state_dict(built_lib, weights0):
    from roma_amd import roma_model
    sd, dsd = weights0
    bad = dict(sd)
    bad.pop("decoder.gps.16.pos_conv.bias")
    with pytest.raises(RuntimeError, match="missing key"):  # strict load, roma_models.py:204
        roma_model((112, 112), False, device="cuda:0", weights=bad, dinov2_weights=dsd, amp_dtype=torch.float32)
    bad = dict(sd)
    bad["extra.key"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="unexpected key"):
        roma_model((112, 112), False, device="cuda:0", weights=bad, dinov2_weights=dsd, amp_dtype=torch.float32)


def test_small_configs_vs_reference_golden(built_lib):
    """224 -> 336, B=2: non-symmetric full and symmetric coarse-only (other seeds)."""
    from roma_amd import roma_model, synthetic
    g = np.load(os.path.join(GOLDEN, "match_small.npz"))
    sd, dsd = synthetic.make_matcher_state_dict(3), synthetic.make_dinov2_state_dict(3)
    inp = _to_dev(synthetic.make_inputs(2, 224, 336, seed=4))
    m = roma_model((224, 224), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=torch.float32,
                   symmetric=False, upsample_res=(336, 336), max_batch=2)
    warp, cert = m.match(inp["im_A"], inp["im_B"], im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
    dw = np.abs(warp.cpu().numpy()[:, ::3, ::3] - g["nonsym_warp"]).max()
    dc = np.abs(cert.cpu().numpy()[:, ::3, ::3] - g["nonsym_cert"]).max()
    print(f"small nonsym: {dw:.3e} {dc:.3e}")
    assert dw < TOL and dc < TOL
    m.symmetric, m.upsample_preds = True, False
    warp, cert = m.match(inp["im_A"], inp["im_B"])
    dw = np.abs(warp.cpu().numpy()[:, ::3, ::3] - g["coarse_warp"]).max()
    dc = np.abs(cert.cpu().numpy()[:, ::3, ::3] - g["coarse_cert"]).max()
    print(f"small coarse sym: {dw:.3e} {dc:.3e}")
    assert dw < TOL and dc < TOL


def test_full_resolution_vs_reference_golden(built_lib, weights0):
    """BASELINE config 5 geometry (560 -> 864, fp32) at B=1 against the reference's own output (sub-sampled 1/8)."""
    from roma_amd import roma_model, synthetic
    g = np.load(os.path.join(GOLDEN, "match_full.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "match_full.json")))
    sd, dsd = weights0
    inp = _to_dev(synthetic.make_inputs(1, 560, 864, seed=1))
    m = roma_model((560, 560), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=torch.float32,
                   symmetric=True, upsample_res=(864, 864), max_batch=1)
    m.debug = True
    warp, cert = m.match(inp["im_A"], inp["im_B"], im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
    torch.cuda.synchronize()
    w, c = warp.cpu().numpy(), cert.cpu().numpy()
    logits = m.debug_fetch("logits16").reshape(2, 1600, 4104)
    am = logits[:, :, :4096].argmax(-1).reshape(2, 40, 40)
    flips = int((am != g["cls16_argmax"]).sum())
    ew = np.abs(w[:, ::8, ::8] - g["warp_sub"])
    ec = np.abs(c[:, ::8, ::8] - g["cert_sub"])
    print(f"full: max|dwarp|={ew.max():.3e} max|dcert|={ec.max():.3e} argmax flips={flips} (oracle min top-2 gap {meta['min_top2_gap']:.2e})")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(max_dwarp=float(ew.max()), max_dcert=float(ec.max()), flips=flips,
                   frac_warp_over=float((ew > TOL).mean()), frac_cert_over=float((ec > TOL).mean())),
              open("gpurun_out/full_parity.json", "w"))
    assert flips == 0, "coarse arg-max flipped w.r.t. the reference (knife-edge logits; see SURVEY hard part 1)"
    assert ew.max() < TOL and ec.max() < TOL
    # whole-tensor checksums (size-independent property): row sums of the reference output
    assert np.allclose(w.sum(axis=(2, 3), dtype=np.float64), g["warp_rowsum"], atol=0.05)
    assert np.allclose(c.sum(axis=2, dtype=np.float64), g["cert_rowsum"], atol=0.05)


@pytest.mark.parametrize("amp", [torch.float32, torch.bfloat16])
def test_stream_split_matches_single_stream(built_lib, weights0, amp):
    """Batches of >= 2 pairs as sub-batches on two HIP streams (model.h `streams`).  Pairs are independent everywhere in
    match(), so the split (here 3 pairs -> 2 + 1, own arenas, fork / join events) must reproduce the single-stream
    result BIT FOR BIT, also from a caller stream that is not the default one.  (Round 1 saw 1-5 % of the bf16 runs
    differ by ~1 bf16 ulp inside a small patch; since the round-2 GEMM epilogue rewrite 0 of 2 700 stress runs do, while
    the library of the commit before it still deviates on the same box - DESIGN.md section 4, profiles/r02_*stress*.)"""
    from roma_amd import roma_model, synthetic
    sd, dsd = weights0
    inp = synthetic.make_inputs(3, 112, 168, seed=7)
    d = _to_dev(inp)
    m = roma_model((112, 112), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=amp,
                   symmetric=True, upsample_res=(168, 168), max_batch=3)
    kw = dict(im_A_high_res=d["im_A_high_res"], im_B_high_res=d["im_B_high_res"])
    m.dual_stream = False
    w1, c1 = m.match(d["im_A"], d["im_B"], **kw)
    m.dual_stream = True
    s = torch.cuda.Stream()
    for it in range(4):
        if it < 3:
            w2, c2 = m.match(d["im_A"], d["im_B"], **kw)
        else:  # a caller stream other than the default one
            with torch.cuda.stream(s):
                w2, c2 = m.match(d["im_A"], d["im_B"], **kw)
            s.synchronize()
        assert torch.equal(w1, w2) and torch.equal(c1, c2), (it, float((c1 - c2).abs().max()), float((w1 - w2).abs().max()))


@pytest.mark.parametrize("fuse", [1, 0])
def test_stream_split_bit_identical_1000_runs(built_lib, weights0, fuse):
    """The acceptance stress for the two-stream split in the benchmarked (bf16) mode: 1 000 runs per refiner-block form
    (fused dw5x5 + 1x1 kernel / separate kernels - the form that deviated 5x more often in round 1), every one
    bit-identical to the single-stream result of the same handle.  ~15 s each."""
    from roma_amd import _lib, roma_model, synthetic
    sd, dsd = weights0
    d = _to_dev(synthetic.make_inputs(3, 112, 168, seed=7))
    m = roma_model((112, 112), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=torch.bfloat16,
                   symmetric=True, upsample_res=(168, 168), max_batch=3)
    _lib.check(_lib.load().roma_set_option(m._handle, b"fuse_refiner_blocks", fuse))
    kw = dict(im_A_high_res=d["im_A_high_res"], im_B_high_res=d["im_B_high_res"])
    m.dual_stream = False
    w1, c1 = m.match(d["im_A"], d["im_B"], **kw)
    m.dual_stream = True
    bad = []
    for it in range(1000):
        w2, c2 = m.match(d["im_A"], d["im_B"], **kw)
        if not (torch.equal(w1, w2) and torch.equal(c1, c2)):
            bad.append((it, float((c1 - c2).abs().max()), float((w1 - w2).abs().max())))
    assert not bad, (len(bad), bad[:5])


def test_handle_cache_keeps_two_resolutions(built_lib, weights0):
    """Tensors of another resolution than the configured one get their own library handle (matcher.py:822-826 only warns);
    the matcher keeps the last two, so alternating between two resolutions rebuilds nothing and results repeat exactly."""
    import warnings
    from roma_amd import roma_model, synthetic
    sd, dsd = weights0
    m = roma_model((112, 112), False, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=torch.float32,
                   symmetric=True, max_batch=1)
    a = _to_dev(synthetic.make_inputs(1, (112, 112), None, seed=7))
    b = _to_dev(synthetic.make_inputs(1, (112, 168), None, seed=8))
    c = _to_dev(synthetic.make_inputs(1, (168, 112), None, seed=9))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wa0, _ = m.match(a["im_A"], a["im_B"])
        h_a = m._handle.value
        wb0, _ = m.match(b["im_A"], b["im_B"])
        h_b = m._handle.value
        assert h_a != h_b and len(m._cache) == 2
        for _ in range(2):  # alternate: both handles are reused, nothing is rebuilt
            wa, _ = m.match(a["im_A"], a["im_B"])
            assert m._handle.value == h_a and torch.equal(wa, wa0)
            wb, _ = m.match(b["im_A"], b["im_B"])
            assert m._handle.value == h_b and torch.equal(wb, wb0)
        m.match(c["im_A"], c["im_B"])  # a third configuration evicts the least recently used one (a)
        assert len(m._cache) == 2 and h_b in [h.value for h in m._cache.values()]
        wa, _ = m.match(a["im_A"], a["im_B"])  # rebuilt: same result
        assert torch.equal(wa, wa0)


def test_non_square_and_pil_inputs(built_lib, weights0, tmp_path):
    """Non-square resolutions (different token grid h != w) and the path / PIL input route (matcher.py:806-816, 853-868)."""
    from PIL import Image
    from oracle import roma_oracle as O
    from roma_amd import roma_model, synthetic
    sd, dsd = weights0
    m = roma_model((112, 168), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=torch.float32,
                   symmetric=True, upsample_res=(160, 240), max_batch=1)
    inp = synthetic.make_inputs(1, (112, 168), (160, 240), seed=5)
    d = _to_dev(inp)
    warp, cert = m.match(d["im_A"], d["im_B"], im_A_high_res=d["im_A_high_res"], im_B_high_res=d["im_B_high_res"])
    w_ref, c_ref = O.match(inp["im_A"], inp["im_B"], sd, dsd, inp["im_A_high_res"], inp["im_B_high_res"])
    assert warp.shape == (1, 160, 480, 4)
    assert (warp.cpu() - w_ref).abs().max() < TOL and (cert.cpu() - c_ref).abs().max() < TOL
    # the PIL and the path route are one route (B = 1, _check_input; matcher.py:530-547)
    g = np.random.Generator(np.random.PCG64(3))
    ims = [Image.fromarray(g.integers(0, 255, size=(90, 130, 3), dtype=np.uint8), "RGB") for _ in range(2)]
    pa, pb = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    ims[0].save(pa)
    ims[1].save(pb)
    w1, c1 = m.match(ims[0], ims[1])
    w2, c2 = m.match(pa, pb)
    assert torch.equal(w1, w2) and torch.equal(c1, c2)
    with pytest.raises(NotImplementedError):  # utils.py:659-661
        m.match(ims[0].convert("L"), ims[1])
    # tensors whose high-resolution size is not upsample_res: the reference fails with a RuntimeError (shape mismatch,
    # matcher.py:891-894) and never rewrites the attribute - neither do we
    bad = _to_dev(synthetic.make_inputs(1, (112, 168), (168, 252), seed=5))
    with pytest.raises(RuntimeError):
        m.match(d["im_A"], d["im_B"], im_A_high_res=bad["im_A_high_res"], im_B_high_res=bad["im_B_high_res"])
    assert tuple(m.upsample_res) == (160, 240)


@pytest.mark.parametrize("tag,coarse,up", [("sq", (112, 112), (168, 168)), ("rect", (112, 140), (168, 196))])
def test_match_from_paths_vs_reference_golden(built_lib, weights0, tag, coarse, up):
    """SURVEY 8a row a2, pinned on the REFERENCE: tests/golden/match_path.npz holds the unmodified reference's own
    match(path, path) on the demo pair (tests/golden/pair_{A,B}.png = the decoded pixels of assets/sacre_coeur_{A,B}.jpg;
    tools/make_goldens.py assets / match_path): _check_input, the coarse transform (matcher.py:806-816), the second transform
    at the upsample resolution (matcher.py:853-868), B = 1.  f32 at the north-star tolerance; the PIL route gives the same
    bits as the path route; `upsample_preds = False` afterwards (a mutable attribute, README.md:82-90) gives the reference's
    coarse-only result from the same paths."""
    from PIL import Image
    from roma_amd import roma_model
    sd, dsd = weights0
    g = np.load(os.path.join(GOLDEN, "match_path.npz"))
    pa, pb = os.path.join(GOLDEN, "pair_A.png"), os.path.join(GOLDEN, "pair_B.png")
    m = roma_model(coarse, True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=torch.float32,
                   symmetric=True, upsample_res=up, max_batch=1)
    warp, cert = m.match(pa, pb)
    torch.cuda.synchronize()
    assert tuple(warp.shape) == g[f"{tag}_warp"].shape and tuple(cert.shape) == g[f"{tag}_cert"].shape
    dw = np.abs(warp.cpu().numpy() - g[f"{tag}_warp"]).max()
    dc = np.abs(cert.cpu().numpy() - g[f"{tag}_cert"]).max()
    print(f"match(path, path) {tag}: max|dwarp|={dw:.3e} max|dcert|={dc:.3e}")
    assert dw < TOL and dc < TOL
    w2, c2 = m.match(Image.open(pa).convert("RGB"), Image.open(pb).convert("RGB"))
    assert torch.equal(warp, w2) and torch.equal(cert, c2)
    from pathlib import Path
    w3, _ = m.match(Path(pa), Path(pb))  # os.PathLike as well as str
    assert torch.equal(warp, w3)
    if tag == "sq":
        m.upsample_preds = False
        wc, cc = m.match(pa, pb)
        assert tuple(wc.shape) == g["sq_coarse_only_warp"].shape
        assert np.abs(wc.cpu().numpy() - g["sq_coarse_only_warp"]).max() < TOL
        assert np.abs(cc.cpu().numpy() - g["sq_coarse_only_cert"]).max() < TOL


def test_forward_apis_vs_reference_golden(tiny_model):
    """RegressionMatcher.forward / forward_symmetric / extract_backbone_features (matcher.py:585-596, 631-670) through
    roma_forward, against the unmodified reference's per-scale `corresps` (tests/golden/match_forward.npz,
    tools/make_goldens.py forward): the symmetric coarse pass with match()'s scale factor, the upsample pass seeded with its
    finest correspondences (matcher.py:870-889), the non-symmetric forward at B = 2 with the default scale_factor = 1, and the
    feature pyramid.  f32; the keys of every corresps[s] are exactly the reference's eval-mode keys."""
    import math
    from roma_amd import synthetic
    g = np.load(os.path.join(GOLDEN, "match_forward.npz"))
    m = tiny_model
    d = _to_dev(synthetic.make_inputs(1, 112, 168, seed=1))
    worst = {}

    def chk(tag, cor, scales):
        assert sorted(cor.keys()) == sorted(scales)
        for s_ in scales:
            assert set(cor[s_].keys()) == {"certainty", "flow"}
            ef = float((cor[s_]["flow"].cpu() - torch.from_numpy(g[f"{tag}_flow{s_}"])).abs().max())
            ec = float((cor[s_]["certainty"].cpu() - torch.from_numpy(g[f"{tag}_cert{s_}"])).abs().max())
            assert tuple(cor[s_]["flow"].shape) == g[f"{tag}_flow{s_}"].shape and tuple(cor[s_]["certainty"].shape) == g[f"{tag}_cert{s_}"].shape
            worst[f"{tag}{s_}"] = (ef, ec)
            assert ef < TOL and ec < TOL, (tag, s_, ef, ec)

    cor = m.forward_symmetric({"im_A": d["im_A"], "im_B": d["im_B"]}, scale_factor=math.sqrt(112 * 112 / 560 ** 2))
    chk("sym", cor, [16, 8, 4, 2, 1])
    cu = m.forward_symmetric({"im_A": d["im_A_high_res"], "im_B": d["im_B_high_res"], "corresps": cor[1]}, upsample=True,
                             batched=True, scale_factor=math.sqrt(168 * 168 / 560 ** 2))
    chk("up", cu, [8, 4, 2, 1])
    d2 = _to_dev(synthetic.make_inputs(2, 112, None, seed=7))
    chk("fwd", m.forward({"im_A": d2["im_A"], "im_B": d2["im_B"]}), [16, 8, 4, 2, 1])
    print("forward APIs, max |d flow| / |d certainty| per scale:", {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in worst.items()})
    # match() == the two passes chained by hand + the epilogue's own tests: the finest upsample flow is what match() clamps
    warp, _ = m.match(d["im_A"], d["im_B"], im_A_high_res=d["im_A_high_res"], im_B_high_res=d["im_B_high_res"])
    a2b = cu[1]["flow"][:1].permute(0, 2, 3, 1).clamp(-1, 1)
    assert float((warp[:, :, :168, 2:] - a2b).abs().max()) < 1e-6
    fp = m.extract_backbone_features({"im_A": d["im_A"], "im_B": d["im_B"]})
    sub = {16: 1, 8: 1, 4: 2, 2: 2, 1: 4}
    assert sorted(fp.keys()) == [1, 2, 4, 8, 16]
    for s_, f in fp.items():
        assert tuple(f.shape) == tuple(g[f"feat{s_}_shape"])
        ref = torch.from_numpy(g[f"feat{s_}"])
        err = float((f[:, :, ::sub[s_], ::sub[s_]].float().cpu() - ref).abs().max())
        assert err < 1e-4 * max(1.0, float(ref.abs().max())), (s_, err)
    fu = m.extract_backbone_features({"im_A": d["im_A_high_res"], "im_B": d["im_B_high_res"]}, upsample=True)
    assert sorted(fu.keys()) == list(g["feat_up_scales"])
    assert float((fu[8].float().cpu() - torch.from_numpy(g["feat_up8"])).abs().max()) < 1e-4 * max(1.0, float(np.abs(g["feat_up8"]).max()))
    with pytest.raises(ValueError):
        m.forward({"im_A": d["im_A_high_res"], "im_B": d["im_B_high_res"]}, upsample=True)  # no batch["corresps"]


def test_composed_out_conv_equals_the_two_step_evaluation(tiny_model):
    """The last ConvRefiner block ends in a 1x1 convolution and is followed directly by out_conv (matcher.py:92-122, 175-178):
    two linear maps with nothing in between.  The library evaluates them as ONE C -> 3 map composed at roma_finalize (wide
    scales; option "compose_out_conv", default on) - 1/9 of the refiners' 1x1 GEMM work disappears.  In f32 the composed and the
    two-step evaluation agree to rounding (and both sit at 1e-6 of the reference golden, test_tiny_match_vs_reference_golden)."""
    from roma_amd import synthetic
    m = tiny_model
    d = _to_dev(synthetic.make_inputs(1, 112, 168, seed=1))
    kw = dict(im_A_high_res=d["im_A_high_res"], im_B_high_res=d["im_B_high_res"])
    assert m.compose_out_conv
    w1, c1 = m.match(d["im_A"], d["im_B"], **kw)
    m.compose_out_conv = False
    try:
        w0, c0 = m.match(d["im_A"], d["im_B"], **kw)
    finally:
        m.compose_out_conv = True
    dw, dc = float((w1 - w0).abs().max()), float((c1 - c0).abs().max())
    print(f"composed vs two-step out_conv (f32): max|dwarp| = {dw:.2e}, max|dcert| = {dc:.2e}")
    assert dw < 2e-6 and dc < 2e-5
    assert not torch.equal(c1, c0)  # the switch really selects another evaluation order
