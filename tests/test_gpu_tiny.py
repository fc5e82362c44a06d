"""Tiny RoMa on the device (roma_amd.TinyRoMa: csrc/tiny.hip + the GEMM / convolution kernels) against the reference's own
TinyRoMa output (tests/golden/tiny_reference.npz, made by tools/make_goldens.py tinyroma with the seeded stand-in XFeat
backbone; tests/golden/tiny_xfeat_reference.npz with a backbone of the real XFeat layer list).  The backbone itself runs
on the device too (forward_single replayed through roma_op_gray_instnorm / conv2d_nhwc / avgpool_nhwc / resize_bilinear /
add3): no torch arithmetic anywhere in match().  fp32, north-star tolerance 1e-3 max-abs on warp and certainty."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _model():
    from roma_amd import TinyRoMa, synthetic
    return TinyRoMa(xfeat=synthetic.XFeatStandIn(0), weights=synthetic.make_tiny_state_dict(0), device="cuda:0")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_tiny_from_reference_features(built_lib, tag):
    """Everything after the backbone (correlation volume, soft arg-max embedding, both matchers, up-sampling) from the
    features the reference itself computed: coarse and fine correspondences against the reference's."""
    g = np.load(os.path.join(GOLDEN, "tiny_reference.npz"))
    m = _model()
    n = g[tag + "_im_A"].shape[0]
    ff, fc = torch.from_numpy(g[tag + "_feat_fine"]).cuda(), torch.from_numpy(g[tag + "_feat_coarse"]).cuda()
    H1, W1 = (g[tag + "_im_B"].shape[-2] // 32) * 32, (g[tag + "_im_B"].shape[-1] // 32) * 32
    cor = m.forward_from_features(ff[:n], fc[:n], ff[n:], fc[n:], H1, W1)
    for lvl in (8, 4):
        df = float((cor[lvl]["flow"].cpu() - torch.from_numpy(g[f"{tag}_flow{lvl}"])).abs().max())
        dc = float((cor[lvl]["certainty"].cpu() - torch.from_numpy(g[f"{tag}_cert{lvl}"])).abs().max())
        print(f"tiny {tag} level {lvl}: max|dflow| = {df:.2e}, max|dcert logit| = {dc:.2e}")
        assert df < TOL and dc < TOL, (tag, lvl, df, dc)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_tiny_match_end_to_end(built_lib, tag):
    """TinyRoMa.match with the stand-in backbone replayed on the device: warp and certainty at the resolution of image A;
    'b' (100 x 150) exercises the resize to multiples of 32."""
    g = np.load(os.path.join(GOLDEN, "tiny_reference.npz"))
    m = _model()
    a, b = torch.from_numpy(g[tag + "_im_A"]).cuda(), torch.from_numpy(g[tag + "_im_B"]).cuda()
    warp, cert = m.match(a, b)
    assert warp.shape == g[tag + "_warp"].shape and cert.shape == g[tag + "_cert"].shape
    dw = float((warp.cpu() - torch.from_numpy(g[tag + "_warp"])).abs().max())
    dc = float((cert.cpu() - torch.from_numpy(g[tag + "_cert"])).abs().max())
    print(f"tiny match {tag}: max|dwarp| = {dw:.2e}, max|dcert| = {dc:.2e}")
    assert dw < TOL and dc < TOL
    w1, c1 = m.match(a[:1], b[:1], batched=True)  # pairs are independent and every kernel is batch-size invariant
    assert torch.equal(w1, warp[:1]) and torch.equal(c1, cert[:1])


@pytest.mark.parametrize("tag,exact", [("x", False), ("e", True), ("y", False)])
def test_tiny_xfeat_architecture_backbone_on_device(built_lib, tag, exact):
    """Backbone of the real XFeat architecture (BasicLayer stacks with BatchNorm, stride-2 and 1 x 1 layers, AvgPool skip)
    replayed on the device: forward_single's two feature maps against the reference's, then both correspondence levels and
    match(); 'e' = the exact_softmax=True branch of pos_embed (tiny.py:139-141), 'y' = 100 x 150 (resize to multiples of 32)."""
    from roma_amd import TinyRoMa, synthetic
    g = np.load(os.path.join(GOLDEN, "tiny_xfeat_reference.npz"))
    m = TinyRoMa(xfeat=synthetic.XFeatArch(0), weights=synthetic.make_tiny_state_dict(0), device="cuda:0", exact_softmax=exact)
    a, b = torch.from_numpy(g[tag + "_im_A"]).cuda(), torch.from_numpy(g[tag + "_im_B"]).cuda()
    xa, rh, rw = m.preprocess_tensor(a)
    xb, _, _ = m.preprocess_tensor(b)
    assert xa.shape[-2] % 32 == 0 and xa.shape[-1] % 32 == 0 and abs(rh - a.shape[-2] / xa.shape[-2]) < 1e-12
    fine, coarse = m.forward_single(torch.cat([xa, xb], dim=0))
    ef = float((fine.cpu() - torch.from_numpy(g[tag + "_feat_fine"])).abs().max())
    ec = float((coarse.cpu() - torch.from_numpy(g[tag + "_feat_coarse"])).abs().max())
    sc = float(torch.from_numpy(g[tag + "_feat_coarse"]).abs().max())
    print(f"xfeat-arch backbone {tag}: max|d fine| = {ef:.2e}, max|d coarse| = {ec:.2e} (range {sc:.1f})")
    assert ef < 1e-4 * max(1.0, sc) and ec < 1e-4 * max(1.0, sc)
    cor = m.forward({"im_A": a, "im_B": b})
    for lvl in (8, 4):
        df = float((cor[lvl]["flow"].cpu() - torch.from_numpy(g[f"{tag}_flow{lvl}"])).abs().max())
        dc = float((cor[lvl]["certainty"].cpu() - torch.from_numpy(g[f"{tag}_cert{lvl}"])).abs().max())
        assert df < TOL and dc < TOL, (tag, lvl, df, dc)
    warp, cert = m.match(a, b)
    dw = float((warp.cpu() - torch.from_numpy(g[tag + "_warp"])).abs().max())
    dc = float((cert.cpu() - torch.from_numpy(g[tag + "_cert"])).abs().max())
    print(f"xfeat-arch match {tag}: max|dwarp| = {dw:.2e}, max|dcert| = {dc:.2e}")
    assert dw < TOL and dc < TOL


def test_tiny_backbone_refuses_layers_it_cannot_replay(built_lib):
    """No silent fallback to the caller's torch module: an unsupported layer is an error at construction."""
    from roma_amd import TinyRoMa, synthetic
    xf = synthetic.XFeatStandIn(0)
    xf.block2 = torch.nn.Sequential(torch.nn.Conv2d(24, 24, 3, padding=1), torch.nn.GELU())
    with pytest.raises(NotImplementedError):
        TinyRoMa(xfeat=xf, weights=synthetic.make_tiny_state_dict(0), device="cuda:0")


def test_tiny_match_demo_size(built_lib):
    """A 480 x 640 pair (the size class of the reference's demo assets, BASELINE config 1): 4 800 x 4 800 correlation volume
    per pair on the MFMA GEMM; outputs against the reference's, 1/4 sub-sampled in the golden."""
    from roma_amd import synthetic
    g = np.load(os.path.join(GOLDEN, "tiny_reference.npz"))
    m = _model()
    inp = synthetic.make_tiny_inputs(1, 480, 640, seed=int(g["c_seed"][0]))
    warp, cert = m.match(inp["im_A"].cuda(), inp["im_B"].cuda())
    dw = float((warp[:, ::4, ::4].cpu() - torch.from_numpy(g["c_warp_sub"])).abs().max())
    dc = float((cert[:, ::4, ::4].cpu() - torch.from_numpy(g["c_cert_sub"])).abs().max())
    print(f"tiny match 480 x 640: max|dwarp| = {dw:.2e}, max|dcert| = {dc:.2e}")
    assert dw < TOL and dc < TOL


@pytest.mark.parametrize("tag,exact", [("p", False), ("pe", True)])
def test_tiny_match_from_path_on_the_demo_pair(built_lib, tag, exact):
    """BASELINE config 1 as the reference runs it: match(path, path) -> match_from_path (tiny.py:193-198) on the demo pair
    (tests/golden/pair_{A,B}.png = the decoded pixels of assets/sacre_coeur_{A,B}.jpg).  A is 480 x 640, B is 640 x 618 ->
    640 x 608: the two images go through forward_single separately (tiny.py:288-290), the correlation volume is 4 800 x 6 080.
    Against the reference's own run (tests/golden/tiny_path_reference.npz: 1/4 sub-sample + per-row sums of the full output)."""
    from roma_amd import TinyRoMa, synthetic
    g = np.load(os.path.join(GOLDEN, "tiny_path_reference.npz"))
    m = TinyRoMa(xfeat=synthetic.XFeatArch(0), weights=synthetic.make_tiny_state_dict(0), device="cuda:0", exact_softmax=exact)
    pa, pb = os.path.join(GOLDEN, "pair_A.png"), os.path.join(GOLDEN, "pair_B.png")
    warp, cert = m.match(pa, pb)  # str -> match_from_path -> un-batched outputs
    assert tuple(warp.shape) == tuple(g[f"{tag}_shape"]) == (480, 640, 4) and tuple(cert.shape) == (480, 640)
    dw = float((warp[::4, ::4].cpu() - torch.from_numpy(g[f"{tag}_warp_sub"])).abs().max())
    dc = float((cert[::4, ::4].cpu() - torch.from_numpy(g[f"{tag}_cert_sub"])).abs().max())
    rs = float((warp.double().sum(dim=(0 + 1, 2)).float().cpu() - torch.from_numpy(g[f"{tag}_warp_rowsum"])).abs().max())
    print(f"tiny match(path, path) {tag}: max|dwarp| = {dw:.2e}, max|dcert| = {dc:.2e}, max|d row sum| = {rs:.2e}")
    assert dw < TOL and dc < TOL and rs < 640 * 4 * TOL
    from pathlib import Path
    w2, c2 = m.match(Path(pa), Path(pb))
    assert torch.equal(warp, w2) and torch.equal(cert, c2)


def test_tiny_rejects_cpu_and_missing_backbone(built_lib):
    from roma_amd import TinyRoMa, synthetic, tiny_roma_v1_outdoor
    with pytest.raises(Exception):
        TinyRoMa(xfeat=synthetic.XFeatStandIn(0), weights=synthetic.make_tiny_state_dict(0), device="cpu")
    with pytest.raises(ValueError):
        tiny_roma_v1_outdoor("cuda:0")
    sd = synthetic.make_tiny_state_dict(0)
    del sd["fine_matcher.4.bias"]
    with pytest.raises(RuntimeError):
        TinyRoMa(xfeat=synthetic.XFeatStandIn(0), weights=sd, device="cuda:0")
