"""Tiny RoMa on the device (roma_amd.TinyRoMa: csrc/tiny.hip + the GEMM / convolution kernels) against the reference's own
TinyRoMa output (tests/golden/tiny_reference.npz, made by tools/make_goldens.py tinyroma with the seeded stand-in XFeat
backbone).  fp32, north-star tolerance 1e-3 max-abs on warp and certainty."""
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
    """TinyRoMa.match with the stand-in backbone running as the caller's torch module on the GPU (its convolutions are
    not ours): warp and certainty at the resolution of image A; 'b' (100 x 150) exercises the resize to multiples of 32."""
    g = np.load(os.path.join(GOLDEN, "tiny_reference.npz"))
    m = _model()
    a, b = torch.from_numpy(g[tag + "_im_A"]).cuda(), torch.from_numpy(g[tag + "_im_B"]).cuda()
    warp, cert = m.match(a, b)
    assert warp.shape == g[tag + "_warp"].shape and cert.shape == g[tag + "_cert"].shape
    dw = float((warp.cpu() - torch.from_numpy(g[tag + "_warp"])).abs().max())
    dc = float((cert.cpu() - torch.from_numpy(g[tag + "_cert"])).abs().max())
    print(f"tiny match {tag}: max|dwarp| = {dw:.2e}, max|dcert| = {dc:.2e}")
    assert dw < TOL and dc < TOL
    w1, c1 = m.match(a[:1], b[:1], batched=True)  # pairs are independent (the torch backbone may pick another conv algorithm
    assert float((w1 - warp[:1]).abs().max()) < 1e-4 and float((c1 - cert[:1]).abs().max()) < 1e-4  # for another batch size)


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
