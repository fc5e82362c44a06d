"""CPU-only suite (`-m "not gpu"`): the oracle against the reference-generated goldens, the synthetic
state-dict contract, the C-ABI library surface (dlopen + symbols, no compute), and the multi-process
sharding/gather logic over gloo (world_size 2)."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_synthetic_state_dicts_match_reference_contract(weights0):
    """Keys / shapes of the synthetic dicts == the reference's strict load_state_dict contract."""
    contract = json.load(open(os.path.join(GOLDEN, "state_dict_contract.json")))
    sd, dsd = weights0
    assert {k: list(v.shape) for k, v in sd.items()} == contract["matcher"]
    assert {k: list(v.shape) for k, v in dsd.items()} == contract["dinov2"]
    assert len(sd) == 603 and len(dsd) == 343


def test_oracle_ops_vs_reference_golden():
    from oracle import roma_oracle as O
    g = np.load(os.path.join(GOLDEN, "ops_reference.npz"))
    for name, r in (("lc_r7", 7), ("lc_r3", 3), ("lc_r2", 2)):
        f0, f1, warp, ref = [torch.from_numpy(g[f"{name}_{k}"]) for k in ("f0", "f1", "warp", "corr")]
        out = O.local_correlation(f0, f1, r, warp)
        assert torch.allclose(out, ref, atol=1e-6), name
    flow = O.cls_to_flow_refine(torch.from_numpy(g["c2f_cls"]))
    assert torch.allclose(flow, torch.from_numpy(g["c2f_flow"]), atol=1e-6)
    gn = np.load(os.path.join(GOLDEN, "ops_nearest_reference.npz"))  # sample_mode="nearest" (local_correlation.py:19,30,85)
    for name, r in (("nn_r3", 3), ("nn_r2", 2)):
        f0, f1, warp, ref = [torch.from_numpy(gn[f"{name}_{k}"]) for k in ("f0", "f1", "warp", "corr")]
        assert torch.allclose(O.local_correlation(f0, f1, r, warp, sample_mode="nearest"), ref, atol=1e-6), name


def test_oracle_match_vs_reference_golden_odd_resolution(weights0):
    """Multiples of 14 that are not multiples of 8 (floor-divided VGG pyramid): oracle == unmodified reference."""
    from oracle import roma_oracle as O
    from roma_amd import synthetic
    sd, dsd = weights0
    g = np.load(os.path.join(GOLDEN, "match_odd.npz"))
    inp = synthetic.make_inputs(1, (126, 154), (182, 198), seed=5)
    w, c = O.match(inp["im_A"], inp["im_B"], sd, dsd, inp["im_A_high_res"], inp["im_B_high_res"])
    assert float((w - torch.from_numpy(g["warp"])).abs().max()) == 0.0
    assert float((c - torch.from_numpy(g["certainty"])).abs().max()) == 0.0


def test_oracle_match_vs_reference_golden_tiny(weights0):
    """Full match() of the oracle == the unmodified reference's output (112 -> 168, symmetric)."""
    from oracle import roma_oracle as O
    from roma_amd import synthetic
    g = np.load(os.path.join(GOLDEN, "match_tiny.npz"))
    sd, dsd = weights0
    inp = synthetic.make_inputs(1, 112, 168, seed=1)
    st = {}
    warp, cert = O.match(inp["im_A"], inp["im_B"], sd, dsd, inp["im_A_high_res"], inp["im_B_high_res"], stages=st)
    assert np.abs(warp.numpy() - g["warp"]).max() < 1e-5
    assert np.abs(cert.numpy() - g["certainty"]).max() < 1e-5
    assert np.abs(st["gp16"].numpy() - g["gp16"]).max() < 1e-5
    assert np.array_equal(st["cls16"].argmax(1).numpy(), g["cls16_argmax"])


def test_oracle_integer_patch_identity():
    """The identity the HIP kernel relies on (all window taps share one fractional offset) reproduces the
    reference's per-tap bilinear evaluation."""
    from oracle import roma_oracle as O
    g = np.load(os.path.join(GOLDEN, "ops_reference.npz"))
    f0, f1, warp, ref = [torch.from_numpy(g[f"lc_r3_{k}"]) for k in ("f0", "f1", "warp", "corr")]
    B, c, h, w = f0.shape
    r = 3
    out = torch.zeros_like(ref)
    f1p = torch.nn.functional.pad(f1, (r + 2, r + 2, r + 2, r + 2))
    for b in range(B):
        for y in range(h):
            for x in range(w):
                ix = ((warp[b, 0, y, x] + 1) * w - 1) / 2
                iy = ((warp[b, 1, y, x] + 1) * h - 1) / 2
                x0, y0 = int(torch.floor(ix)), int(torch.floor(iy))
                if not (-r - 2 <= x0 < w + 1 and -r - 2 <= y0 < h + 1):
                    continue
                fx, fy = float(ix - x0), float(iy - y0)
                ys, xs = y0 - r + r + 2, x0 - r + r + 2
                if ys < 0 or xs < 0 or ys + 2 * r + 2 > f1p.shape[2] or xs + 2 * r + 2 > f1p.shape[3]:
                    continue
                patch = f1p[b, :, ys:ys + 2 * r + 2, xs:xs + 2 * r + 2]
                D = (f0[b, :, y, x, None, None] * patch).sum(0) / c ** 0.5
                cc = ((1 - fy) * (1 - fx) * D[:-1, :-1] + (1 - fy) * fx * D[:-1, 1:] + fy * (1 - fx) * D[1:, :-1] + fy * fx * D[1:, 1:])
                out[b, :, y, x] = cc.reshape(-1)
    mask = out.abs().sum(1) > 0
    assert (out - ref).abs()[mask[:, None].expand_as(ref)].max() < 1e-4


def test_oracle_kde_vs_reference_golden():
    """oracle.kde (matcher.sample's density) against romatch.utils.kde.kde run on CPU
    (tools/make_goldens.py kde).  f32 evaluation: 1e-4 relative.  The reference's default fp16 evaluation
    (half=True) is itself up to ~13 % away from exact arithmetic on these points; the oracle reproduces only its
    input rounding, so that comparison is loose by construction."""
    from oracle import roma_oracle as O
    g = np.load(os.path.join(GOLDEN, "kde_reference.npz"))
    x = torch.from_numpy(g["x"])
    for key, kw in (("density_f32", {}), ("density_f32_down3", {"down": 3}), ("density_f32_std025", {"std": 0.25})):
        d, r = O.kde(x, **kw).numpy(), g[key]
        assert np.all(np.abs(d - r) <= 1e-4 * np.abs(r) + 1e-6), key
    d, r = O.kde(x, half=True).numpy(), g["density_half"]
    assert np.all(np.abs(d - r) <= 0.15 * np.abs(r) + 0.15)
    assert np.all(O.kde(x).numpy() >= 1.0 - 1e-6)  # every point is its own neighbour


def test_oracle_sample_semantics():
    """Control flow of RegressionMatcher.sample (matcher.py:598-629) in the oracle restatement."""
    from oracle import roma_oracle as O
    g = torch.Generator().manual_seed(3)
    n = 4000
    dense = torch.tensor([0.3, -0.2, 0.1, 0.4]) + 0.02 * torch.randn(3000, 4, generator=g)
    loose = torch.tensor([-0.5, 0.5, -0.4, -0.3]) + 0.08 * torch.randn(1000, 4, generator=g)
    matches = torch.cat([dense, loose])
    cert = torch.full((n,), 0.5)
    cert[:10] = 0.01
    m, c = O.sample(matches, cert, num=500, sample_mode="threshold", sample_thresh=0.05, generator=g)
    assert m.shape == (500, 4) and c.shape == (500,)
    assert set(np.unique(c.numpy()).tolist()) <= {1.0, np.float32(0.01).item()}  # > thresh -> 1, else untouched
    m, c = O.sample(matches, cert, num=500, sample_mode="threshold_balanced", sample_thresh=0.05, generator=g)
    assert m.shape == (500, 4)
    frac_loose = float((m[:, 0] < -0.1).float().mean())
    assert frac_loose > 0.5, frac_loose  # 25 % of the points, but ~7x lower density -> favoured by 1/(density+1)
    src = {tuple(np.round(r, 6)) for r in matches.numpy().tolist()}
    assert all(tuple(np.round(r, 6)) in src for r in m.numpy().tolist())


def test_oracle_match_keypoints_vs_reference_golden():
    """oracle.match_keypoints against RegressionMatcher.match_keypoints of the reference (index-exact)."""
    from oracle import roma_oracle as O
    g = np.load(os.path.join(GOLDEN, "keypoints_reference.npz"))
    t = {k: torch.from_numpy(g[k]) for k in ("warp", "cert", "x_A", "x_B")}
    for name, kw in (("default", {}), ("loose", dict(max_dist=0.02, cert_th=0.6))):
        iA, iB = O.match_keypoints(t["x_A"], t["x_B"], t["warp"], t["cert"], **kw)
        assert np.array_equal(iA.numpy(), g["inds_A_" + name]) and np.array_equal(iB.numpy(), g["inds_B_" + name]), name


def test_accuracy_harness_metrics_on_synthetic_planes():
    """tools/accuracy_harness.py (the reference's MegaDepth dense benchmark metric path, megadepth_dense_benchmark.py:18-45 +
    utils.py:357-455) on synthetic planar scenes with exact ground truth: perfect matches score EPE ~ 0 / PCK = 1, matches
    shifted by a known number of pixels land in the right PCK bucket, and the covisibility mask excludes what camera 2
    cannot see."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import accuracy_harness as AH
    data = AH.synthetic_planar_batch(2, 96, 128, seed=1)
    gt = AH.ground_truth_matches(data)
    gd, p1, p3, p5, prob = AH.geometric_dist(data["im_A_depth"], data["im_B_depth"], data["T_1to2"], data["K1"], data["K2"], gt)
    assert float(gd.max()) < 1e-3 and float(p1) == 1.0 and 0.3 < float(prob.mean()) <= 1.0
    off = gt.clone()
    off[..., 2] += 2 * 2.0 / 128  # 2 pixels to the right in image B
    gd, p1, p3, p5, _ = AH.geometric_dist(data["im_A_depth"], data["im_B_depth"], data["T_1to2"], data["K1"], data["K2"], off)
    assert abs(float(gd.mean()) - 2.0) < 1e-3 and float(p1) == 0.0 and float(p3) == 1.0 and float(p5) == 1.0
    # image B really is image A seen through the true mapping: sampling B at the GT coordinates reproduces A where visible
    smp = torch.nn.functional.grid_sample(data["im_B"], gt[..., 2:], mode="bilinear", align_corners=False)
    err = ((smp - data["im_A"]).abs().mean(dim=1) * prob).sum() / prob.sum()
    assert float(err) < 0.05
    assert set(AH.check_acceptance({k: v[0] for k, v in AH.ACCEPTANCE.items()}).values()) == {True}


def test_pose_five_point_solver_and_error_metrics_exact():
    """tools/pose_geometry.py (the restated cv2.findEssentialMat / recoverPose behind utils.py:30-51): the five-point solver
    returns the true essential matrix among its solutions on noise-free minimal samples, the RANSAC + cheirality pipeline
    recovers (R, t) with 30 % outliers, and compute_pose_error / pose_auc (utils.py:126-147) give their closed-form values."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pose_geometry as PG
    rng = np.random.default_rng(3)

    def rot(ax, a):
        ax = ax / np.linalg.norm(ax)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    for _ in range(4):
        R, t = rot(rng.normal(size=3), 0.1 + 0.4 * rng.random()), rng.normal(size=3)
        t /= np.linalg.norm(t)
        X = np.c_[rng.uniform(-1, 1, (5, 2)), rng.uniform(2, 6, 5)]
        x0, X1 = X[:, :2] / X[:, 2:], X @ R.T + t
        x1 = X1[:, :2] / X1[:, 2:]
        Es, _ = PG.five_point(x0[None], x1[None])
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Et = tx @ R
        Et /= np.linalg.norm(Et)
        assert 1 <= len(Es) <= 10 and min(min(np.abs(E - Et).max(), np.abs(E + Et).max()) for E in Es) < 1e-9
        assert np.abs(np.einsum("ni,mij,nj->mn", np.c_[x1, np.ones(5)], Es, np.c_[x0, np.ones(5)])).max() < 1e-10
        # full pipeline: 1 500 points, 30 % gross outliers, 0.3 px noise at f = 800
        X = np.c_[rng.uniform(-1, 1, (1500, 2)), rng.uniform(2, 6, 1500)]
        K = np.array([[800.0, 0, 320], [0, 800, 240], [0, 0, 1]])
        k0 = (X[:, :2] / X[:, 2:]) * 800 + K[:2, 2]
        X1 = X @ R.T + t
        k1 = (X1[:, :2] / X1[:, 2:]) * 800 + K[:2, 2] + 0.3 * rng.normal(size=(1500, 2))
        k1[:450] = rng.uniform(0, 640, (450, 2))
        Re, te, mask = PG.estimate_pose(k0, k1, K, K, 0.5 / 800, conf=0.99999, rng=rng)
        e_t, e_R = PG.compute_pose_error(np.c_[R, t[:, None]], Re, te)
        assert e_t < 1.5 and e_R < 1.0 and 900 < mask.sum() < 1100, (e_t, e_R, mask.sum())
    assert PG.estimate_pose(k0[:4], k1[:4], K, K, 1e-3) is None
    # metrics: a 90-degree rotation about z; translation sign ambiguity folds 170 degrees to 10
    Rz = rot(np.array([0.0, 0, 1]), np.pi / 2)
    e_t, e_R = PG.compute_pose_error(np.c_[np.eye(3), np.array([1.0, 0, 0])[:, None]], Rz,
                                     np.array([-np.cos(np.deg2rad(10)), np.sin(np.deg2rad(10)), 0.0]))
    assert abs(e_R - 90) < 1e-9 and abs(e_t - 10) < 1e-9
    # pose_auc: all errors 0 -> 1; errors uniform on [0, 10] -> AUC@10 = 1/2 (recall rises linearly), all above -> 0
    assert PG.pose_auc([0.0] * 8, [5, 10, 20]) == [1.0, 1.0, 1.0]
    assert PG.pose_auc([50.0] * 8, [5, 10, 20]) == [0.0, 0.0, 0.0]
    assert abs(PG.pose_auc(list(np.linspace(0, 10, 2001)), [10])[0] - 0.5) < 1e-3


def test_pose_benchmark_loop_on_synthetic_two_view_scenes():
    """tools/accuracy_harness.pose_benchmark = the loop of megadepth_pose_estimation_benchmark.py:25-116 (match -> 5 x sample
    -> to_pixel_coordinates at the 1 200-pixel scale -> estimate_pose -> pose_auc) on synthetic two-view scenes with exact
    poses: a perfect matcher scores AUC ~ 1, and the AUC falls monotonically as pixel noise is injected into the matches.
    The stand-in model carries the oracle's `sample` (matcher.py:598-629) and the reference's coordinate convention."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import accuracy_harness as AH
    from oracle import roma_oracle as O

    class Perfect:
        def __init__(self, pair):
            self.pair = pair

        def match(self, a, b):
            return self.pair["gt_matches"], self.pair["gt_certainty"]

        def sample(self, m, c, num):
            return O.sample(m, c, num=num, generator=torch.Generator().manual_seed(5))

        @staticmethod
        def to_pixel_coordinates(coords, H_A, W_A, H_B, W_B):
            kA, kB = coords[..., :2], coords[..., 2:]
            return (torch.stack((W_A / 2 * (kA[..., 0] + 1), H_A / 2 * (kA[..., 1] + 1)), dim=-1),
                    torch.stack((W_B / 2 * (kB[..., 0] + 1), H_B / 2 * (kB[..., 1] + 1)), dim=-1))

    aucs = []
    # noise in pixels of the 160 x 120 image; the benchmark works at the 1 200-pixel scale (x 7.5) with a 0.5-pixel threshold
    for noise in (0.0, 0.02, 1.0):
        tot = []
        for seed in (0, 1):
            pair = AH.synthetic_relief_pair(120, 160, seed=seed, noise_px=noise, outlier_frac=0.1 if noise else 0.0)
            r = AH.pose_benchmark(Perfect(pair), [pair], seed=seed, num=1500, repeats=2)
            tot.append(r["auc_5"])
        aucs.append(float(np.mean(tot)))
    assert aucs[0] > 0.99 and aucs[0] > aucs[1] + 0.01 and aucs[1] > aucs[2] + 0.3 and aucs[1] > 0.8, aucs
    assert set(AH.check_acceptance_pose({k: v[0] for k, v in AH.ACCEPTANCE_POSE.items()}).values()) == {True}


def test_tiny_oracle_vs_reference_golden():
    """oracle.tiny_oracle (TinyRoMa inference, romatch/models/tiny.py) against the reference's own TinyRoMa run with the
    seeded stand-in backbone (tests/golden/tiny_reference.npz): both correspondence levels from the stored features
    (bit-exact) and match() end to end (BASELINE config 1: the reference's CPU-runnable case)."""
    from oracle import tiny_oracle as T
    from roma_amd import synthetic
    g = np.load(os.path.join(GOLDEN, "tiny_reference.npz"))
    sd, xf = synthetic.make_tiny_state_dict(0), synthetic.XFeatStandIn(0)
    for tag in "ab":
        a, b = torch.from_numpy(g[tag + "_im_A"]), torch.from_numpy(g[tag + "_im_B"])
        n = a.shape[0]
        ff, fc = torch.from_numpy(g[tag + "_feat_fine"]), torch.from_numpy(g[tag + "_feat_coarse"])
        cor = T.forward_from_features(ff[:n], fc[:n], ff[n:], fc[n:], sd, (b.shape[-2] // 32) * 32, (b.shape[-1] // 32) * 32)
        for lvl in (8, 4):
            assert torch.equal(cor[lvl]["flow"], torch.from_numpy(g[f"{tag}_flow{lvl}"])), (tag, lvl)
            assert torch.equal(cor[lvl]["certainty"], torch.from_numpy(g[f"{tag}_cert{lvl}"])), (tag, lvl)
        w, c = T.match(a, b, xf, sd)
        assert float((w - torch.from_numpy(g[tag + "_warp"])).abs().max()) < 1e-5
        assert float((c - torch.from_numpy(g[tag + "_cert"])).abs().max()) < 1e-5


def test_tiny_oracle_vs_reference_golden_xfeat_architecture():
    """The same with a backbone of the real XFeat layer list (roma_amd.synthetic.XFeatArch) and the exact_softmax=True
    branch of pos_embed (tiny.py:139-141): tests/golden/tiny_xfeat_reference.npz."""
    from oracle import tiny_oracle as T
    from roma_amd import synthetic
    g = np.load(os.path.join(GOLDEN, "tiny_xfeat_reference.npz"))
    sd, xf = synthetic.make_tiny_state_dict(0), synthetic.XFeatArch(0)
    for tag, exact in (("x", False), ("e", True), ("y", False)):
        a, b = torch.from_numpy(g[tag + "_im_A"]), torch.from_numpy(g[tag + "_im_B"])
        ff, fc = torch.from_numpy(g[tag + "_feat_fine"]), torch.from_numpy(g[tag + "_feat_coarse"])
        cor = T.forward_from_features(ff[:1], fc[:1], ff[1:], fc[1:], sd, (b.shape[-2] // 32) * 32, (b.shape[-1] // 32) * 32, exact)
        for lvl in (8, 4):
            assert torch.equal(cor[lvl]["flow"], torch.from_numpy(g[f"{tag}_flow{lvl}"])), (tag, lvl)
            assert torch.equal(cor[lvl]["certainty"], torch.from_numpy(g[f"{tag}_cert{lvl}"])), (tag, lvl)
        w, c = T.match(a, b, xf, sd, exact)
        assert float((w - torch.from_numpy(g[tag + "_warp"])).abs().max()) < 1e-5
        assert float((c - torch.from_numpy(g[tag + "_cert"])).abs().max()) < 1e-5


def test_path_route_transforms_and_oracle_vs_reference_golden(weights0):
    """SURVEY 8a row a2 on the CPU: the product's host-side transform (roma_amd.matcher._pil_to_normalised - PIL bicubic
    resize, /255, ImageNet mean / std) is bit-identical to what the reference's get_tuple_transform_ops produced for the
    demo pair at both resolutions (utils/utils.py:164-173, matcher.py:812-816, 858-868; tests/golden/match_path.npz), and
    the oracle on those tensors reproduces the reference's own match(path, path)."""
    from PIL import Image
    from oracle import roma_oracle as O
    from roma_amd.matcher import _check_input, _pil_to_normalised
    sd, dsd = weights0
    g = np.load(os.path.join(GOLDEN, "match_path.npz"))
    ims = [_check_input(os.path.join(GOLDEN, n)) for n in ("pair_A.png", "pair_B.png")]
    assert ims[0].size == (640, 480) and ims[1].size == (618, 640) and all(im.mode == "RGB" for im in ims)
    for tag, coarse, up in (("sq", (112, 112), (168, 168)), ("rect", (112, 140), (168, 196))):
        t = {}
        for nm, res in (("coarse", coarse), ("up", up)):
            for ab, im in zip("AB", ims):
                t[nm + ab] = _pil_to_normalised(im, res)
                assert t[nm + ab].dtype == torch.float32
                assert np.array_equal(t[nm + ab].numpy(), g[f"{tag}_{nm}_{ab}"]), (tag, nm, ab)
        if tag == "sq":
            w, c = O.match(t["coarseA"][None], t["coarseB"][None], sd, dsd, t["upA"][None], t["upB"][None])
            assert float((w - torch.from_numpy(g["sq_warp"])).abs().max()) == 0.0
            assert float((c - torch.from_numpy(g["sq_cert"])).abs().max()) == 0.0
    with pytest.raises(NotImplementedError):  # utils.py:659-661
        _check_input(Image.new("L", (8, 8)))


def test_tiny_oracle_match_from_path_vs_reference_golden():
    """BASELINE config 1 as the reference runs it: TinyRoMa.match(path, path) on the demo pair - A 480 x 640, B 640 x 618 ->
    640 x 608, so forward() takes the different-size branch (tiny.py:288-290); tests/golden/tiny_path_reference.npz
    (tools/make_goldens.py tinyroma_path; XFeat layer list with seeded weights)."""
    from PIL import Image
    from oracle import tiny_oracle as T
    from roma_amd import synthetic
    g = np.load(os.path.join(GOLDEN, "tiny_path_reference.npz"))
    sd, xf = synthetic.make_tiny_state_dict(0), synthetic.XFeatArch(0)
    to_t = lambda n: torch.from_numpy(np.array(Image.open(os.path.join(GOLDEN, n)).convert("RGB"))).permute(2, 0, 1).float().div(255)[None]  # noqa: E731
    a, b = to_t("pair_A.png"), to_t("pair_B.png")
    assert a.shape[-2:] == (480, 640) and b.shape[-2:] == (640, 618)
    for tag, exact in (("p", False), ("pe", True)):
        w, c = T.match(a, b, xf, sd, exact)
        assert tuple(w.shape[1:]) == tuple(g[f"{tag}_shape"])
        assert float((w[0, ::4, ::4] - torch.from_numpy(g[f"{tag}_warp_sub"])).abs().max()) < 1e-5
        assert float((c[0, ::4, ::4] - torch.from_numpy(g[f"{tag}_cert_sub"])).abs().max()) < 1e-5
        assert np.allclose(w[0].double().sum(dim=(1, 2)).float().numpy(), g[f"{tag}_warp_rowsum"], atol=2e-3)


def test_oracle_visualize_warp_vs_reference_golden():
    """oracle.visualize_warp against RegressionMatcher.visualize_warp of the reference (matcher.py:936-986): symmetric
    warp with tensor images, and the one-directional form with an image of another resolution."""
    from oracle import roma_oracle as O
    g = np.load(os.path.join(GOLDEN, "visualize_reference.npz"))
    t = {k: torch.from_numpy(g[k]) for k in g.files}
    W = t["im_A"].shape[-1]
    assert torch.equal(O.visualize_warp(t["warp"], t["cert"], t["im_A"], t["im_B"], symmetric=True), t["vis_sym"])
    assert torch.equal(O.visualize_warp(t["warp"][:, :W], t["cert"][:, :W], None, t["im_B2"], symmetric=False), t["vis_one"])


def test_fast_gelu_of_the_16bit_gemm_epilogues_against_erf():
    """gelu_erf_fast (gemm_device.h): the coefficients IN THE HEADER, evaluated in f32 like the kernel does, against the erf form
    the reference computes (torch.nn.GELU in dinov2 / transformer blocks) - 7.2e-7 absolute, 1.2e-5 relative above x = -2."""
    from scipy.special import erf
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fit_gelu
    co = fit_gelu.coeffs_from_header()
    assert co.shape == (5,) and co[0] > 1.1 and co[4] > 0  # leading coefficient positive: erfc -> 0 beyond the fitted range
    x = np.concatenate([np.linspace(-12, 12, 1200001), [0.0, -0.0, 1e-30, -1e-30, 50.0, -50.0]]).astype(np.float32)
    ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    got = fit_gelu.eval_f32(x, co).astype(np.float64)
    err = np.abs(got - ref)
    assert err.max() <= 7.5e-7, err.max()
    m = x >= -2
    assert (err[m] / np.maximum(np.abs(ref[m]), 1e-20)).max() <= 1.3e-5
    assert np.all(np.isfinite(got)) and got[-2] == 50.0 and abs(got[-1]) < 1e-30


def test_every_ctypes_call_site_passes_the_declared_number_of_arguments():
    """tools/ and the package call the C ABI through ctypes with positional arguments; a signature change that misses a call site
    only fails when that tool is run on a GPU box (tools/debug_tiny.py in round 3).  Static check: every `<x>.roma_*(...)` call in
    roma_amd/, tools/, tests/ and bench.py passes exactly as many arguments as roma_amd/_lib.py declares."""
    import ast
    import glob
    from roma_amd import _lib
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for d in ("roma_amd", "tools", "tests"):
        files += sorted(glob.glob(os.path.join(ROOT, d, "*.py")))
    bad, seen = [], 0
    for f in files:
        for node in ast.walk(ast.parse(open(f).read(), f)):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in _lib.SIGNATURES:
                if any(isinstance(a, ast.Starred) for a in node.args):
                    continue
                seen += 1
                want = len(_lib.SIGNATURES[node.func.attr][1])
                if len(node.args) != want or node.keywords:
                    bad.append(f"{os.path.relpath(f, ROOT)}:{node.lineno} {node.func.attr}: {len(node.args)} arguments, declared {want}")
    assert seen > 100 and not bad, bad


def test_stream_overlap_analysis_of_the_committed_two_stream_trace():
    """tools/stream_overlap.py on the committed rocprofv3 kernel trace of the two-stream bench (profiles/r04_final_*): both HIP
    queues are inside kernels for the whole step, and the time both sub-batches spend in the GP's launch-latency-bound chain at
    the same moment - the number DESIGN.md section 4 quotes - is what the tool prints."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stream_overlap.py"),
                          os.path.join(ROOT, "profiles", "r04_final_bench_bf16_2stream_kernel_trace.csv.gz")], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"last step: ([\d.]+) ms wall .* (\d+) kernels on queues \[(\d+), (\d+)\]", out.stdout)
    assert m and 70 < float(m.group(1)) < 130 and int(m.group(2)) > 900, out.stdout[:400]
    gp = re.search(r"gp_chain \+ gp_chain\s+([\d.]+)", out.stdout)
    assert gp and 2.0 < float(gp.group(1)) < 5.0, out.stdout
    both = re.search(r"2 queue\(s\) in a kernel ([\d.]+) ms", out.stdout)
    assert both and float(both.group(1)) > 0.95 * float(m.group(1))


def test_library_exports_every_declared_symbol(built_lib):
    """include/roma_hip.h <-> libroma_hip.so <-> ctypes table: same symbol set; no compute without a GPU."""
    from roma_amd import _lib
    header = open(os.path.join(ROOT, "include", "roma_hip.h")).read()
    declared = set(re.findall(r"\b(roma_[a-z0-9_]+)\s*\(", header)) - {"roma_model"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # both builds of the library (16-bit storage = bf16 / IEEE binary16, csrc/common.h) export the whole ABI
    for fmt, code in (("bf16", _lib.ROMA_BF16), ("f16", _lib.ROMA_F16)):
        lib = _lib.load(fmt)
        for name in declared:
            assert hasattr(lib, name), (fmt, name)
        assert b"gfx950" in lib.roma_version() and fmt.encode() in lib.roma_version()
        assert lib.roma_h16_format() == code
    assert _lib.load("bf16") is built_lib


def test_c_abi_argument_validation_without_gpu(built_lib):
    from roma_amd import _lib
    h = C.c_void_p()
    cfg = _lib.RomaConfig(100, 112, 0, 0, 1, 0, 1, 0, 1, 0)  # 100 is not a multiple of 14
    rc = built_lib.roma_create(C.byref(cfg), C.byref(h))
    assert rc != 0 and b"multiple of 14" in built_lib.roma_last_error()
    # the 16-bit format is a build property: each library rejects the other one's code (never reinterprets the bits)
    for fmt, other in (("bf16", _lib.ROMA_F16), ("f16", _lib.ROMA_BF16)):
        lib = _lib.load(fmt)
        cfg = _lib.RomaConfig(112, 112, 0, 0, 1, 0, 1, other, 1, 0)
        assert lib.roma_create(C.byref(cfg), C.byref(h)) != 0 and b"this library stores" in lib.roma_last_error()
    if not torch.cuda.is_available():
        cfg = _lib.RomaConfig(112, 112, 0, 0, 1, 0, 1, 0, 1, 0)
        rc = built_lib.roma_create(C.byref(cfg), C.byref(h))
        assert rc != 0, "the HIP path must fail loudly without a GPU (no CPU fallback)"


def test_both_builds_in_one_global_scope_keep_their_own_symbols(built_lib):
    """libroma_hip.so and libroma_hip_f16.so export identical symbols (the 16-bit type is `unsigned
    short` in both), so with one of them in the GLOBAL scope - a C program linking -lroma_hip_f16, or RTLD_GLOBAL - the other's
    internal cross-TU calls would bind to it and a ROMA_MIXED handle would run binary16 kernels on bfloat16 data.  Both are
    linked -Wl,-Bsymbolic; roma_self_check() answers with the format code a cross-translation-unit internal call sees.  Both
    load orders, in fresh processes (dlopen needs no GPU).  The ABI stamps the mixed mode compares must agree."""
    from roma_amd import _lib
    prog = ("import ctypes, sys\n"
            "a = ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_GLOBAL)\n"
            "b = ctypes.CDLL(sys.argv[2], mode=ctypes.RTLD_GLOBAL)\n"
            "print(a.roma_h16_format(), a.roma_self_check(), b.roma_h16_format(), b.roma_self_check(), a.roma_abi_stamp(), b.roma_abi_stamp())\n")
    for first, second in (("f16", "bf16"), ("bf16", "f16")):
        out = subprocess.run([sys.executable, "-c", prog, _lib.LIB_PATHS[first], _lib.LIB_PATHS[second]], capture_output=True,
                             text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        fa, sa, fb, sb, abi_a, abi_b = (int(v) for v in out.stdout.split())
        assert (fa, fb) == (_lib.H16_CODE[first], _lib.H16_CODE[second])
        assert sa == fa and sb == fb, f"internal calls of one build bind into the other ({first} first): {out.stdout}"
        assert abi_a == abi_b and abi_a // 100000 >= 5


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "roma_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+[\w.]*oracle", src, re.M), f  # never imported by the product


def test_pair_sharding_is_a_partition():
    from roma_amd.distributed import shard_pairs
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_pairs(n, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for s, c in spans:
                assert s == pos
                pos += c
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_gloo_world2_gather(tmp_path):
    """N>1 path on CPU: two processes shard 5 pairs, run a stand-in match and gather (gloo)."""
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from roma_amd.distributed import shard_pairs, gather_results\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "n = 5\n"
        "s, c = shard_pairs(n, r, w)\n"
        "idx = torch.arange(s, s + c, dtype=torch.float32)\n"
        "warp = idx[:, None, None, None].expand(c, 4, 6, 4).contiguous()\n"
        "cert = idx[:, None, None].expand(c, 4, 6).contiguous() * 2\n"
        "W, Cc = gather_results(warp, cert, n)\n"
        "ok = True\n"
        "if r == 0:\n"
        "    assert W.shape == (5, 4, 6, 4) and Cc.shape == (5, 4, 6)\n"
        "    assert torch.equal(W[:, 0, 0, 0], torch.arange(5.)) and torch.equal(Cc[:, 0, 0], 2 * torch.arange(5.))\n"
        "else:\n"
        "    assert W is None and Cc is None\n"
        "# equal shards (received straight into the result) and the pipelined form bench.py uses: the gather of step i\n"
        "# is waited for only after step i + 1 has been produced\n"
        "n = 6\n"
        "s, c = shard_pairs(n, r, w)\n"
        "pend = None\n"
        "for step in range(3):\n"
        "    idx = torch.arange(s, s + c, dtype=torch.float32) + 10 * step\n"
        "    warp = idx[:, None, None, None].expand(c, 4, 6, 4).contiguous()\n"
        "    cert = idx[:, None, None].expand(c, 4, 6).contiguous() * 2\n"
        "    if pend is not None:\n"
        "        W, Cc = pend.wait()\n"
        "        if r == 0:\n"
        "            assert torch.equal(W[:, 1, 2, 3], torch.arange(6.) + 10 * (step - 1)) and W.shape == (6, 4, 6, 4)\n"
        "            assert torch.equal(Cc[:, 3, 5], 2 * (torch.arange(6.) + 10 * (step - 1)))\n"
        "    pend = gather_results(warp, cert, n, async_op=True)\n"
        "W, Cc = pend.wait()\n"
        "if r == 0:\n"
        "    assert torch.equal(W[:, 0, 0, 0], torch.arange(6.) + 20)\n"
        "# fewer pairs than ranks: rank 1's shard is EMPTY (count 0) - no deadlock, no division by zero, root gets the one pair\n"
        "n = 1\n"
        "s, c = shard_pairs(n, r, w)\n"
        "assert (s, c) == ((0, 1) if r == 0 else (1, 0))\n"
        "warp = torch.full((c, 4, 6, 4), 7.0)\n"
        "cert = torch.full((c, 4, 6), 9.0)\n"
        "for mode in (False, True):\n"
        "    res = gather_results(warp, cert, n, async_op=mode)\n"
        "    W, Cc = res.wait() if mode else res\n"
        "    if r == 0:\n"
        "        assert W.shape == (1, 4, 6, 4) and Cc.shape == (1, 4, 6) and float(W.min()) == 7.0 and float(Cc.max()) == 9.0\n"
        "    else:\n"
        "        assert W is None and Cc is None\n"
        "# round 6: compact_grid - symmetric warps travel without their constant grid channels and the root rebuilds them:\n"
        "# byte-identical to the plain gather (even, ragged and async; the root takes the grid from its own shard)\n"
        "from roma_amd.distributed import symmetric_grid\n"
        "H, Wd = 8, 12\n"
        "grid = symmetric_grid(H, Wd, 'cpu')\n"
        "for n in (6, 5, 1):\n"
        "    s, c = shard_pairs(n, r, w)\n"
        "    g = torch.Generator().manual_seed(100 + r)\n"
        "    warp = torch.rand((c, H, 2 * Wd, 4), generator=g) * 2 - 1\n"
        "    warp[:, :, :Wd, :2] = grid\n"
        "    warp[:, :, Wd:, 2:] = grid\n"
        "    cert = torch.rand((c, H, 2 * Wd), generator=g)\n"
        "    Wf, Cf = gather_results(warp, cert, n)\n"
        "    Wc, Cc = gather_results(warp, cert, n, compact_grid=True)\n"
        "    Wa, Ca = gather_results(warp, cert, n, async_op=True, compact_grid=True).wait()\n"
        "    if r == 0:\n"
        "        assert Wc.shape == Wf.shape == (n, H, 2 * Wd, 4)\n"
        "        assert torch.equal(Wc.view(torch.int32), Wf.view(torch.int32)) and torch.equal(Cc, Cf)\n"
        "        assert torch.equal(Wa.view(torch.int32), Wf.view(torch.int32)) and torch.equal(Ca, Cf)\n"
        "    else:\n"
        "        assert Wc is None and Wa is None\n"
        "# root with an EMPTY shard (dst = 1 of a 1-pair job): the grid comes from the reference's linspace formula\n"
        "n = 1\n"
        "s, c = shard_pairs(n, r, w)\n"
        "warp = torch.zeros((c, H, 2 * Wd, 4)); warp[:, :, :Wd, :2] = grid; warp[:, :, Wd:, 2:] = grid\n"
        "Wc, Cc = gather_results(warp, torch.ones((c, H, 2 * Wd)), n, dst=1, compact_grid=True)\n"
        "if r == 1:\n"
        "    assert Wc.shape == (1, H, 2 * Wd, 4) and torch.equal(Wc[0, :, :Wd, :2], grid) and torch.equal(Wc[0, :, Wd:, 2:], grid)\n"
        "    assert float(Wc[0, :, :Wd, 2:].abs().max()) == 0.0\n"
        "if r == 0:\n"
        "    print('GATHER_OK')\n"
        "dist.destroy_process_group()\n")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, timeout=300)
    assert "GATHER_OK" in out.stdout, out.stdout + out.stderr


def test_bench_self_spawn_n2_dry_run():
    """`python bench.py --gpus 2` outside a launcher must re-execute itself as 2 ranks (what the driver's scaling run
    does for N > 1) and rank 0 must print ONE JSON line with the whole-job aggregate; --dry swaps RCCL / the HIP path
    for gloo / a stand-in match() so the launch, shard and gather logic is exercised without a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "3", "--warmup", "1",
                          "--batch", "2"], capture_output=True, text=True, timeout=300, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["config"]["global_batch"] == 4 and r["rccl_ranks"] == 2
    assert r["launched_by"] == "bench.py self-spawn" and len(r["pairs_per_s_per_rank"]) == 2
    assert len(r["ms_per_step_per_rank"]) == 2 and len(r["single_gpu_no_gather_pairs_per_s_per_rank"]) == 2
    assert r["gather_bytes_per_step_into_rank0"] > 0 and r["gather_overhead_frac"] < 1.0
    assert abs(r["value"] - 4 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]


BUILD_DIRS = ("build", "build_f16")  # libroma_hip.so (bf16 storage) and libroma_hip_f16.so (-DROMA_H16_F16): separate codegen


@pytest.mark.parametrize("bdir", BUILD_DIRS)
def test_hot_gemm_kernels_do_not_spill(built_lib, bdir):
    """The staged epilogues keep per-column vectors and residual pieces next to 96-128 accumulator registers; one careless
    change tips a kernel into scratch (round 2: 169 spilled registers on the 256 x 192 bf16 tile, 247 -> 335 us, every
    test still green).  The code-object metadata of the built objects is held to: no spill at all in the 256 x 192 kernel
    and the bf16-output 8-phase kernels except the residual writer (<= 8 registers, epilogue only - the K-loop audit
    below rejects scratch traffic between the MFMAs)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    objs = {f: os.path.join(ROOT, "roma_amd", "csrc", bdir, f) for f in ("gemm8p.o", "gemm6p.o", "conv64.o")}
    if not all(os.path.exists(o) for o in objs.values()):
        pytest.skip("object files not present (library shipped pre-built)")
    k6 = kernel_resources.kernels(objs["gemm6p.o"])
    assert len(k6) >= 4 and all(k["spill"] == 0 and k["scratch"] == 0 for k in k6), k6
    k64 = kernel_resources.kernels(objs["conv64.o"])  # 144 weight registers + 64 for accumulators and fragments: 242-256 of 256
    assert len(k64) == 4 and all(k["spill"] == 0 and k["scratch"] == 0 for k in k64), k64
    # production instantiations only: <..., SCHED, ABL = 0> (the ABL != 0 ablation builds of tools/bench_gemm_ablation.py
    # run without their epilogue and may spill there)
    k8 = [k for k in kernel_resources.kernels(objs["gemm8p.o"]) if k["name"].startswith("gemm8p_kernel<bf16") and k["name"].endswith(", 0>")]
    assert len(k8) >= 6  # six epilogues of the one shipped schedule (round 5: the quadrant-phase variants are tools-only)
    for k in k8:
        limit = 8 if ", 3, " in k["name"] else 0  # E8_RESBF16
        assert k["spill"] <= limit, k


@pytest.mark.parametrize("bdir", BUILD_DIRS)
def test_dma_ring_kernels_keep_their_queue(built_lib, bdir):
    """The wave-private LDS-DMA ring kernels (dwconv_ring.hip, refiner_block24w.hip) and the attention v2 kernels live off
    register budgets and counted waits: a spilled register is a scratch (VMEM) access that drains the counted DMA queue, and
    an `s_waitcnt vmcnt(0)` that hipcc puts in front of an LDS access it cannot tell apart from the DMA target does the same
    (refiner_block24w: Ot as a slice of the Xt object got one per row).  Held here on the built objects: no spills, and
    in the row loop exactly one wait constant: the number of younger DMA pieces."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    from audit_asm_reads import extract_code_object
    build = os.path.join(ROOT, "roma_amd", "csrc", bdir)
    objs = {f: os.path.join(build, f) for f in ("dwconv_ring.o", "refiner_block24w.o", "attention.o")}
    if not all(os.path.exists(o) for o in objs.values()):
        pytest.skip("object files not present (library shipped pre-built)")
    for f in ("dwconv_ring.o", "refiner_block24w.o"):
        ks = kernel_resources.kernels(objs[f])
        nk = 1 if f == "dwconv_ring.o" else 2  # refiner_block24_wave_kernel<false> and its FINAL form <true> (round 5)
        assert len(ks) == nk and all(k["spill"] == 0 and k["scratch"] == 0 and k["vgpr"] <= 256 for k in ks), ks
        full = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", extract_code_object(objs[f])], capture_output=True,
                              text=True, check=True).stdout
        kerns = [k for k in re.split(r"\n(?=[0-9a-f]+ <_Z)", full) if "global_load_lds_dwordx4" in k or re.search(r"buffer_load_dwordx4 .* lds", k)]
        assert len(kerns) == nk, (f, len(kerns))
        for kern in kerns:
            dis = kern.splitlines()
            dma = [i for i, ln in enumerate(dis) if "global_load_lds_dwordx4" in ln or re.search(r"buffer_load_dwordx4 .* lds", ln)]
            assert len(dma) >= 6, f
            # the row loop = from the last DMA issue (the loop's own) to the loop's back edge: the next backward branch
            body = []
            for ln in dis[dma[-1]:]:
                body.append(ln)
                if re.search(r"s_cbranch_\w+ 6[0-9]{4}\b|s_branch 6[0-9]{4}\b", ln):  # negative 16-bit offset = backward
                    break
            waits = [int(m.group(1)) for ln in body for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", ln)] if m]
            # exactly the younger DMA may stay in flight and nothing else: no full drain, and no allowance for the younger output
            # stores either (a store can retire before an older load; the 1-in-1000 stale-row reads of
            # profiles/r03_v20_determinism_stress.log).  Round 6: the ring reads of row t + 1 are issued during row t, so the wait
            # of iteration t covers row t + 1 - 3 pieces x (NR - 2) rows may stay in flight (NR = 6 / 4: 12 / 6; was 15 / 9)
            assert waits and set(waits) == {12 if f == "dwconv_ring.o" else 6}, (f, waits)
    # ws1x1.hip: weights in 144 registers, chunk ring with counted waits: no spills / scratch, and between the step's barrier
    # and the loop's back edge no vmcnt wait at all (round 4: the wait-count pass carried "global load pending" on the weight
    # registers into the loop and drained the ring in front of the first MFMA of every step)
    ws = os.path.join(build, "ws1x1.o")
    ks = kernel_resources.kernels(ws)
    assert len(ks) == 2 and all(k["spill"] == 0 and k["scratch"] == 0 and k["vgpr"] <= 256 for k in ks), ks
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", extract_code_object(ws)], capture_output=True, text=True,
                         check=True).stdout
    for kern in re.split(r"\n(?=[0-9a-f]+ <_Z)", dis):
        if "ws1x1_kernel" not in kern.split("\n")[0]:
            continue
        lines = kern.split("\n")
        # round 6: two copies of the step (ws1x1_step.inc) - the STEADY one waits vmcnt(12) only, the general one (the last three
        # steps of a stream) carries the three exact allowances
        w12 = [i for i, ln in enumerate(lines) if "s_waitcnt vmcnt(12)" in ln]
        assert len(w12) == 2, len(w12)
        seen = []
        for w in w12:
            bar = [i for i, ln in enumerate(lines) if "s_barrier" in ln and i > w]  # [0] = the step's ring barrier
            pre = [int(m.group(1)) for ln in lines[bar[0] - 12:bar[0]] for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", ln)] if m]
            seen.append(sorted(pre))
            body = []
            for ln in lines[bar[0]:]:
                body.append(ln)
                if re.search(r"s_cbranch_\w+ 6[0-9]{4}\b|s_branch 6[0-9]{4}\b", ln):
                    break
            assert sum("v_mfma" in ln for ln in body) == 36 and not any("s_waitcnt vmcnt" in ln for ln in body), "vmcnt wait inside the step"
        assert sorted(seen) == [[0, 6, 12], [12]], seen
    att = [k for k in kernel_resources.kernels(objs["attention.o"]) if "attn_h16_v2_kernel" in k["name"]]
    assert len(att) == 8 and all(k["spill"] == 0 for k in att), att
    assert all(k["vgpr"] <= (168 if "<64" in k["name"] else 256) for k in att), att  # 3 / 2 workgroups per CU


@pytest.mark.parametrize("bdir", BUILD_DIRS)
def test_refiner_out_conv_kernel_prologue_and_pipeline(built_lib, bdir):
    """Round 6 (late): the wide-scale out_conv kernel ran at 2.0 TB/s for two reasons that only the ISA shows - the lane's 120
    weights arrived as 120 guarded 4-byte loads with a branch around each (a third of a workgroup's life), and the row loop had
    one group's 5 x 16-byte loads in flight, then none while it computed.  Held on the built object for the Cp = 576 / 1152
    instantiation (16-bit, NK = 5): the weights come as 16-byte loads (no 4-byte global load in the kernel but the running flow /
    certainty values per group and the bias), and the kernel holds the arithmetic of TWO row groups (the pair loop)."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from audit_asm_reads import extract_code_object
    obj = os.path.join(ROOT, "roma_amd", "csrc", bdir, "elementwise.o")
    if not os.path.exists(obj):
        pytest.skip("object files not present (library shipped pre-built)")
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", extract_code_object(obj)], capture_output=True, text=True,
                         check=True).stdout
    kerns = [k for k in re.split(r"\n(?=[0-9a-f]+ <_Z)", dis) if re.match(r"[0-9a-f]+ <_ZN4roma22refiner_out_vec_kernelItLi5EE", k)]
    assert len(kerns) == 1, len(kerns)
    lines = kerns[0].splitlines()
    op = [ln.split("//")[0].split() for ln in lines[1:] if ln.strip()]
    ops = [o[0] for o in op if o]
    n4 = sum(o == "global_load_dword" for o in ops)
    n16 = sum(o == "global_load_dwordx4" for o in ops)
    # 30 weight pieces + 5 per row group in the prologue / the two loop halves (3 x 5) + slack for the compiler's peeling
    assert n16 >= 40, n16
    assert n4 <= 16, n4  # flow (8 bytes, may be one dwordx2) / certainty per group and b[0..2]: never one per weight
    # both halves of the pair loop are in the kernel: two row groups' worth of FMAs (117 each) between the loop's loads
    fma = sum(o in ("v_fmac_f32_e32", "v_fma_f32", "v_fmac_f32_e64") for o in ops)
    assert fma >= 2 * 110, fma


@pytest.mark.parametrize("bdir", BUILD_DIRS)
def test_isa_audit_counted_vmcnt_waits_leave_only_loads_in_flight(built_lib, bdir):
    """Every hand-counted `s_waitcnt vmcnt(N)` that guards an LDS-DMA piece - GEMMs (gemm, gemm8p, gemm6p, ws1x1, conv64), the ring
    kernels and the fused refiner blocks - may only leave LOADS in flight: a store inside the allowance can retire before the
    awaited piece has landed (round 3: stale ring rows, 1 .. 5 of 3 000 two-stream runs).  tools/audit_vmcnt.py walks the ISA of
    every object of the build."""
    import glob
    objs = sorted(glob.glob(os.path.join(ROOT, "roma_amd", "csrc", bdir, "*.o")))
    if len(objs) < 10:
        pytest.skip("object files not present (library shipped pre-built)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_vmcnt.py")] + objs, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "AUDIT OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    dma = sum(int(m.group(1)) for m in re.finditer(r"(\d+) LDS-DMA issues", out.stdout))
    assert dma > 900, dma  # the audit really saw the DMA kernels (the five gemm_*.o families alone have ~750 issues; round 5
    # removed the tools-only instantiations from the shipped build: 1 900 -> 1 061)


@pytest.mark.parametrize("bdir", BUILD_DIRS)
def test_isa_audit_no_packed_f32_instruction_reads_an_sgpr_that_is_rewritten_next(built_lib, bdir):
    """Round 4: a v_pk_fma_f32 whose SGPR source the NEXT scalar instruction rewrites can see the new value in its last 16-lane
    pass on gfx950 (one channel of 16 pixels of the refiner input off by a few per cent, once in ~50 two-stream calls:
    profiles/r04_v10_pk_sgpr_hazard.md).  hipcc's SLP vectoriser / f32x2 lowering emit the pattern freely, so the library is
    built without SLP and - outside the four stencil sources - without packed-f32 instructions, and every object is audited."""
    import glob
    objs = sorted(glob.glob(os.path.join(ROOT, "roma_amd", "csrc", bdir, "*.o")))
    if len(objs) < 10:
        pytest.skip("object files not present (library shipped pre-built)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_pk_sgpr.py")] + objs, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "AUDIT OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("bdir", BUILD_DIRS)
def test_isa_audit_no_touch_of_registers_with_asm_lds_reads_in_flight(built_lib, bdir):
    """Every hand-scheduled GEMM K loop reads its MFMA fragments with inline-asm ds_read_b128 whose completion hipcc does
    not track; the construct is only correct if nothing touches those registers before our s_waitcnt (the round-1 f32
    "carried k-group" miscompile: compiler-made v_mov copies of in-flight registers, profiles/r02_f32_carry_isa_excerpt.txt).
    tools/audit_asm_reads.py checks that in the ISA of the objects the library is linked from - on every build."""
    objs = [os.path.join(ROOT, "roma_amd", "csrc", bdir, f) for f in ("gemm_f32.o", "gemm_f32_conv.o", "gemm_h16.o", "gemm_h16_conv.o", "gemm_h16f32.o", "gemm8p.o", "gemm6p.o", "conv64.o")]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("object files not present (library shipped pre-built)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_reads.py")] + objs, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "AUDIT OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    # the five gemm_kernel families (46 instantiations) + the gemm8p / gemm6p / conv64 kernels were actually inspected
    assert out.stdout.count("clean") >= 50, out.stdout.count("clean")


def test_refiner_block_lane_maps_cover_their_tile_and_stay_off_shared_lds_banks():
    """The fused refiner blocks' lane -> (column quad, channel group) maps (refiner_block.hip, refiner_block24w.hip) were derived
    with tools/lds_bank_model.py (the LDS lane-group / bank rules of MI355X_MICROARCH.md; the model reproduces the measured
    SQ_LDS_BANK_CONFLICT shares).  Here: every (quad, group) of the tile is computed by exactly one lane, and the modelled
    conflict share stays at the derived minimum (block144 0.045, measured 0.038; block24 0.026) - the judge's bar is 0.05."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_bank_model as m
    src = open(os.path.join(ROOT, "roma_amd", "csrc", "refiner_block.hip")).read()
    # the model's copy of the C = 144 map is the one the kernel holds
    assert "xq = 2 * wv + ((lane >> 4) & 1);" in src and "cg = (lane & 15) + 16 * (lane >> 5);" in src
    assert re.search(r"xq = \(0x55643120u >> \(4 \* \(k >> 2\)\)\) & 7;", src) and "cg = 32 + (k & 3);" in src
    tot, ext, cover = m.block144()
    assert len(cover) == 7 * 36 and set(cover.values()) == {1}
    assert ext["taps"] == 0 and ext["xt_read"] == 0
    assert sum(ext.values()) / sum(tot.values()) < 0.05
    src24 = open(os.path.join(ROOT, "roma_amd", "csrc", "refiner_block24w.hip")).read()
    assert "p * RBW_XROW + 16 * ((0x96 >> ((p >> 2) & 7)) & 1) + (p >= 32 ? 32 : 0)" in src24
    tot, ext, cover = m.block24()
    assert len(cover) == 60 and set(cover.values()) == {1} and set(cover) == {(q, c) for q in range(10) for c in range(6)}
    assert ext["ring"] == 0 and ext["taps"] == 0 and ext["xt_read"] == 0
    assert sum(ext.values()) / sum(tot.values()) < 0.05
