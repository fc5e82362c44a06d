"""Accuracy harness for the dense matchers: the reference's two MegaDepth benchmarks wired to roma_amd -
dense PCK (romatch/benchmarks/megadepth_dense_benchmark.py:9-116, acceptance numbers tests/test_mega_dense.py:17-21) and
MegaDepth-1500 pose AUC (romatch/benchmarks/megadepth_pose_estimation_benchmark.py:25-116, acceptance numbers
tests/test_mega1500.py:17-21; `estimate_pose` / `pose_auc` restated in tools/pose_geometry.py because OpenCV is absent).

There is no MegaDepth data and there are no trained weights on this machine, so what runs offline is the whole harness on
SYNTHETIC planar scenes with exact ground truth (a textured plane seen from two cameras: depth maps, intrinsics and the
relative pose are analytic), with the same metric code path the real data takes:

    python tools/accuracy_harness.py --synthetic 4                 # GPU box: plumbing check with seeded random weights
    python tools/accuracy_harness.py --synthetic-pose 2            # GPU box: match -> 5 x sample -> estimate_pose -> AUC
    python tools/accuracy_harness.py --megadepth data/megadepth --weights roma_outdoor.pth --dinov2 dinov2_vitl14_pretrain.pth

With random weights the numbers are meaningless (the matcher has not learnt anything); with the released weights the
MegaDepth run must reproduce ACCEPTANCE below within the reference's own tolerances.  Metric functions work on CPU or GPU
tensors; only `benchmark()` calls the matcher."""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# tests/test_mega_dense.py:17-21 (roma_outdoor, coarse_res 560, symmetric False, upsample_preds False, h = w = 560)
ACCEPTANCE = {"epe": (1.581197752074192, 1e-1), "mega_pck_1": (0.8516846923828125, 2e-3), "mega_pck_3": (0.9566336059570313, 2e-3),
              "mega_pck_5": (0.9714825439453125, 2e-3)}


# tests/test_mega1500.py:17-21 (roma_outdoor, coarse_res 672, upsample_res 1344): (value, atol)
ACCEPTANCE_POSE = {"auc_5": (0.6271474434923545, 3e-1 / 100), "auc_10": (0.7673889435429945, 2e-1 / 100),
                   "auc_20": (0.8642099162282599, 1e-1 / 100)}


def warp_kpts(kpts0, depth0, depth1, T_0to1, K0, K1, relative_depth_error_threshold=0.05):
    """romatch/utils/utils.py:357-455 (bilinear depth interpolation, hard mask): kpts0 [N, L, 2] normalised -> (valid [N, L],
    warped [N, L, 2] normalised in image 1) with the covisibility and relative-depth-consistency checks."""
    n, h, w = depth0.shape
    d0 = F.grid_sample(depth0[:, None], kpts0[:, :, None], mode="bilinear", align_corners=False)[:, 0, :, 0]
    k0 = torch.stack((w * (kpts0[..., 0] + 1) / 2, h * (kpts0[..., 1] + 1) / 2), dim=-1)
    nonzero = d0 != 0
    k0h = torch.cat([k0, torch.ones_like(k0[:, :, [0]])], dim=-1) * d0[..., None]
    cam = K0.inverse() @ k0h.transpose(2, 1)
    wcam = T_0to1[:, :3, :3] @ cam + T_0to1[:, :3, [3]]
    zc = wcam[:, 2, :]
    wh = (K1 @ wcam).transpose(2, 1)
    wk = wh[:, :, :2] / (wh[:, :, [2]] + 1e-4)
    h1, w1 = depth1.shape[1:3]
    covis = (wk[:, :, 0] > 0) * (wk[:, :, 0] < w1 - 1) * (wk[:, :, 1] > 0) * (wk[:, :, 1] < h1 - 1)
    wk = torch.stack((2 * wk[..., 0] / w1 - 1, 2 * wk[..., 1] / h1 - 1), dim=-1)
    d1 = F.grid_sample(depth1[:, None], wk[:, :, None], mode="bilinear", align_corners=False)[:, 0, :, 0]
    consistent = ((d1 - zc) / d1).abs() < relative_depth_error_threshold
    return nonzero * covis * consistent, wk


def geometric_dist(depth1, depth2, T_1to2, K1, K2, dense_matches):
    """megadepth_dense_benchmark.py:18-45: pixel distance of the predicted B-coordinates to the depth-warped A-grid."""
    b, h1, w1, _ = dense_matches.shape
    x1 = dense_matches[..., :2].reshape(b, h1 * w1, 2)
    mask, x2 = warp_kpts(x1.double(), depth1.double(), depth2.double(), T_1to2.double(), K1.double(), K2.double())
    x2 = torch.stack((w1 * (x2[..., 0] + 1) / 2, h1 * (x2[..., 1] + 1) / 2), dim=-1)
    prob = mask.float().reshape(b, h1, w1)
    x2_hat = dense_matches[..., 2:]
    x2_hat = torch.stack((w1 * (x2_hat[..., 0] + 1) / 2, h1 * (x2_hat[..., 1] + 1) / 2), dim=-1)
    gd = (x2_hat - x2.reshape(b, h1, w1, 2)).norm(dim=-1)
    gd = gd[prob == 1]
    return gd, (gd < 1.0).float().mean(), (gd < 3.0).float().mean(), (gd < 5.0).float().mean(), prob


def benchmark(model, batches):
    """megadepth_dense_benchmark.py:47-116 over an iterable of batches {im_A, im_B, im_A_depth, im_B_depth, T_1to2, K1, K2}
    (what MegadepthBuilder's test_loftr split yields); `model` = roma_amd.RegressionMatcher with symmetric=False,
    upsample_preds=False like the reference test."""
    tot = {"epe": 0.0, "mega_pck_1": 0.0, "mega_pck_3": 0.0, "mega_pck_5": 0.0}
    n = 0
    for data in batches:
        dev = model.device if hasattr(model, "device") else "cuda:0"
        d = {k: v.to(dev) for k, v in data.items()}
        matches, _ = model.match(d["im_A"], d["im_B"], batched=True)
        gd, p1, p3, p5, _ = geometric_dist(d["im_A_depth"], d["im_B_depth"], d["T_1to2"], d["K1"], d["K2"], matches)
        tot["epe"] += float(gd.mean()); tot["mega_pck_1"] += float(p1); tot["mega_pck_3"] += float(p3); tot["mega_pck_5"] += float(p5)
        n += 1
    return {k: v / max(n, 1) for k, v in tot.items()}


def check_acceptance(results):
    return {k: abs(results[k] - ref) <= tol for k, (ref, tol) in ACCEPTANCE.items()}


def synthetic_planar_batch(batch, h, w, seed=0):
    """A textured fronto-parallel plane at depth 2 seen by camera 1, camera 2 = camera 1 rotated about the optical axis,
    panned a little and moved sideways / forward: depth maps, intrinsics and T_1to2 are exact, image B is image A's texture
    rendered through the true mapping (so a perfect matcher scores PCK = 1)."""
    g = torch.Generator().manual_seed(seed)
    f = 0.9 * w
    K = torch.tensor([[f, 0.0, w / 2], [0.0, f, h / 2], [0.0, 0.0, 1.0]])
    ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
    out = {k: [] for k in ("im_A", "im_B", "im_A_depth", "im_B_depth", "T_1to2", "K1", "K2")}
    mean = torch.tensor([0.485, 0.456, 0.406])[:, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    for _ in range(batch):
        tex = F.interpolate(torch.rand(1, 3, h // 8, w // 8, generator=g), size=(h, w), mode="bicubic", align_corners=False)[0].clamp(0, 1)
        a = (torch.rand(1, generator=g).item() - 0.5) * 0.2                      # roll
        pan = (torch.rand(1, generator=g).item() - 0.5) * 0.1                    # yaw
        R = torch.tensor([[torch.cos(torch.tensor(a)), -torch.sin(torch.tensor(a)), 0.0],
                          [torch.sin(torch.tensor(a)), torch.cos(torch.tensor(a)), 0.0], [0.0, 0.0, 1.0]])
        Ry = torch.tensor([[torch.cos(torch.tensor(pan)), 0.0, torch.sin(torch.tensor(pan))], [0.0, 1.0, 0.0],
                           [-torch.sin(torch.tensor(pan)), 0.0, torch.cos(torch.tensor(pan))]])
        R = Ry @ R
        t = torch.tensor([(torch.rand(1, generator=g).item() - 0.5) * 0.3, (torch.rand(1, generator=g).item() - 0.5) * 0.2, -0.1])
        T = torch.cat([R, t[:, None]], dim=1)
        # camera-2 depth of the plane z1 = 2 and the inverse mapping 2 -> 1 (to render image B from A's texture)
        nrm, dpl = torch.tensor([0.0, 0.0, 1.0]), 2.0                            # plane n . X1 = d in camera 1
        Hm = K @ (R + t[:, None] @ nrm[None] / dpl) @ K.inverse()                # homography 1 -> 2
        Hi = Hm.inverse()
        p2 = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1) @ Hi.T           # pixel of image 1 seen at each pixel of image 2
        p1 = p2[..., :2] / p2[..., 2:]
        gridn = torch.stack((2 * p1[..., 0] / w - 1, 2 * p1[..., 1] / h - 1), dim=-1)
        imB = F.grid_sample(tex[None], gridn[None], mode="bilinear", align_corners=False)[0]
        # depth of the plane along camera 2's rays: X2 = R X1 + t with n . X1 = d  =>  z2 = d' / (n2 . K^-1 [u, v, 1])
        n2 = R @ nrm
        d2 = dpl + float(n2 @ t)
        rays = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1) @ K.inverse().T
        depthB = d2 / (rays @ n2)
        out["im_A"].append((tex - mean) / std); out["im_B"].append((imB - mean) / std)
        out["im_A_depth"].append(torch.full((h, w), dpl)); out["im_B_depth"].append(depthB)
        out["T_1to2"].append(T); out["K1"].append(K); out["K2"].append(K)
    return {k: torch.stack(v).float() for k, v in out.items()}


def ground_truth_matches(data):
    """Dense matches [B, H, W, 4] = (A grid, exact B coordinates) of a synthetic batch: what a perfect matcher returns."""
    b, h, w = data["im_A_depth"].shape
    ys, xs = torch.meshgrid(torch.linspace(-1 + 1 / h, 1 - 1 / h, h), torch.linspace(-1 + 1 / w, 1 - 1 / w, w), indexing="ij")
    grid = torch.stack((xs, ys), dim=-1)[None].expand(b, h, w, 2)
    _, x2 = warp_kpts(grid.reshape(b, h * w, 2).double(), data["im_A_depth"].double(), data["im_B_depth"].double(),
                      data["T_1to2"].double(), data["K1"].double(), data["K2"].double())
    return torch.cat((grid, x2.reshape(b, h, w, 2).float()), dim=-1)


# ------------------------------------------------------------------------------------------- MegaDepth-1500 pose benchmark
def pose_benchmark(model, pairs, seed=0, num=5000, repeats=5, max_side=1200):
    """megadepth_pose_estimation_benchmark.py:25-116 over an iterable of pairs
    {im_A, im_B (what model.match takes: paths, PIL images or [3, H, W] tensors), K1, K2 [3, 3], T_1to2 [3, 4] or [4, 4],
    size_A = (w1, h1), size_B = (w2, h2) of the ORIGINAL images}: per pair one `match`, then `repeats` x {`sample` 5 000
    matches, pixel coordinates at the 1 200-pixel scale, shuffle, `estimate_pose` at 0.5 px / mean focal, pose error};
    failures count as 90 degrees.  Returns the reference's dictionary (auc_5/10/20, map_5/10/20)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pose_geometry as PG
    rng = np.random.default_rng(seed)
    tot_e_pose = []
    for pair in pairs:
        K1, K2 = np.array(pair["K1"], dtype=np.float64), np.array(pair["K2"], dtype=np.float64)
        T = np.array(pair["T_1to2"], dtype=np.float64)
        R, t = T[:3, :3], T[:3, 3]
        dense_matches, dense_certainty = model.match(pair["im_A"], pair["im_B"])
        (w1, h1), (w2, h2) = pair["size_A"], pair["size_B"]
        s1, s2 = max_side / max(w1, h1), max_side / max(w2, h2)   # the scaling of the DKM / RoMa papers (:58-65)
        w1, h1, w2, h2 = s1 * w1, s1 * h1, s2 * w2, s2 * h2
        K1, K2 = K1.copy(), K2.copy()
        K1[:2] *= s1
        K2[:2] *= s2
        for _ in range(repeats):
            sparse, _ = model.sample(dense_matches, dense_certainty, num)
            k1, k2 = model.to_pixel_coordinates(sparse, h1, w1, h2, w2)
            k1, k2 = k1.detach().cpu().double().numpy(), k2.detach().cpu().double().numpy()
            sh = rng.permutation(len(k1))
            k1, k2 = k1[sh], k2[sh]
            try:
                norm_threshold = 0.5 / (np.mean(np.abs(K1[:2, :2])) + np.mean(np.abs(K2[:2, :2])))
                R_est, t_est, _ = PG.estimate_pose(k1, k2, K1, K2, norm_threshold, conf=0.99999, rng=rng)
                e_t, e_R = PG.compute_pose_error(np.concatenate((R_est, t_est), axis=-1), R, t)
            except Exception as e:  # estimate_pose returned None (too few matches / no model), like the reference's except
                print(repr(e))
                e_t, e_R = 90, 90
            tot_e_pose.append(max(e_t, e_R))
    tot = np.array(tot_e_pose)
    auc = PG.pose_auc(tot, [5, 10, 20])
    acc = [(tot < th).mean() for th in (5, 10, 15, 20)]
    return {"auc_5": auc[0], "auc_10": auc[1], "auc_20": auc[2], "map_5": acc[0], "map_10": float(np.mean(acc[:2])),
            "map_20": float(np.mean(acc))}


def check_acceptance_pose(results):
    return {k: abs(results[k] - ref) <= tol for k, (ref, tol) in ACCEPTANCE_POSE.items()}


def synthetic_relief_pair(h, w, seed=0, noise_px=0.0, outlier_frac=0.0):
    """A two-view scene with EXACT pose and dense correspondences for the pose harness: camera 1 sees a smooth relief
    (depth 3 .. 5, not planar), camera 2 is rotated by a few degrees and translated.  Returns the pair dictionary of
    pose_benchmark plus `gt_matches` [h, w, 4] (normalised A grid, exact B coordinates, optionally perturbed by Gaussian
    pixel noise / a fraction of uniform outliers) and `gt_certainty` [h, w] (1 where the point projects inside image B)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda: torch.rand(1, generator=g).item()
    f = 0.9 * w
    K = torch.tensor([[f, 0.0, w / 2], [0.0, f, h / 2], [0.0, 0.0, 1.0]], dtype=torch.float64)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64) + 0.5, torch.arange(w, dtype=torch.float64) + 0.5, indexing="ij")
    depth = 4.0 + 0.6 * torch.sin(xs / w * 5.0 + 6.0 * r()) + 0.4 * torch.cos(ys / h * 4.0 + 6.0 * r())
    ax = torch.tensor([r() - 0.5, r() - 0.5, 0.3 * (r() - 0.5)], dtype=torch.float64)
    ang = 0.05 + 0.1 * r()
    ax = ax / ax.norm()
    Kx = torch.tensor([[0.0, -ax[2], ax[1]], [ax[2], 0.0, -ax[0]], [-ax[1], ax[0], 0.0]], dtype=torch.float64)
    R = torch.eye(3, dtype=torch.float64) + torch.sin(torch.tensor(ang)) * Kx + (1 - torch.cos(torch.tensor(ang))) * Kx @ Kx
    t = torch.tensor([0.5 * (r() - 0.5) + 0.3, 0.3 * (r() - 0.5), 0.2 * (r() - 0.5)], dtype=torch.float64)
    pix = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1)
    X1 = (pix @ K.inverse().T) * depth[..., None]
    X2 = X1 @ R.T + t
    p2 = X2 @ K.T
    p2 = p2[..., :2] / p2[..., 2:]
    vis = (p2[..., 0] > 0) & (p2[..., 0] < w) & (p2[..., 1] > 0) & (p2[..., 1] < h) & (X2[..., 2] > 0)
    if noise_px > 0:
        p2 = p2 + noise_px * torch.randn(p2.shape, generator=g, dtype=torch.float64)
    if outlier_frac > 0:
        bad = torch.rand(h, w, generator=g) < outlier_frac
        rnd = torch.rand(h, w, 2, generator=g, dtype=torch.float64) * torch.tensor([w, h], dtype=torch.float64)
        p2 = torch.where(bad[..., None], rnd, p2)
    gridA = torch.stack((2 * xs / w - 1, 2 * ys / h - 1), dim=-1)
    gridB = torch.stack((2 * p2[..., 0] / w - 1, 2 * p2[..., 1] / h - 1), dim=-1)
    return {"im_A": None, "im_B": None, "K1": K.numpy(), "K2": K.numpy(), "T_1to2": torch.cat([R, t[:, None]], dim=1).numpy(),
            "size_A": (w, h), "size_B": (w, h), "gt_matches": torch.cat((gridA, gridB), dim=-1).float(),
            "gt_certainty": vis.float()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic planar batches (of 2 pairs) to run")
    ap.add_argument("--synthetic-pose", type=int, default=0, help="number of synthetic pairs for the pose (MegaDepth-1500) loop")
    ap.add_argument("--megadepth", default=None, help="data root of the MegaDepth test split (reference: data/megadepth)")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--dinov2", default=None)
    ap.add_argument("--res", type=int, default=560)
    args = ap.parse_args()
    from roma_amd import roma_outdoor, synthetic
    if args.weights:
        sd, dsd = torch.load(args.weights, map_location="cpu"), torch.load(args.dinov2, map_location="cpu")
    else:
        sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)
    model = roma_outdoor(device="cuda:0", weights=sd, dinov2_weights=dsd, coarse_res=args.res, symmetric=False, upsample_preds=False,
                         max_batch=2)
    if args.megadepth:
        if not os.path.isdir(args.megadepth):
            raise FileNotFoundError(f"{args.megadepth}: the MegaDepth test_loftr split (scene_info npz + images + depths) is not on "
                                    "this machine; see romatch/datasets/megadepth.py for the expected layout")
        raise NotImplementedError("MegaDepth loader: feed benchmark() with batches of the keys listed in its docstring "
                                  "(romatch.datasets.MegadepthBuilder.build_scenes(split='test_loftr', ht=res, wt=res))")
    if args.synthetic_pose:
        # the pose loop end to end through roma_amd's match + sample (planar synthetic images; with random weights the matches
        # are noise, every pose fails or is wrong and the AUC is ~0: a plumbing check of the harness, not a result)
        pairs = []
        for s in range(args.synthetic_pose):
            d = synthetic_planar_batch(1, args.res, args.res, seed=100 + s)
            pairs.append({"im_A": d["im_A"].to("cuda:0"), "im_B": d["im_B"].to("cuda:0"), "K1": d["K1"][0].numpy(),
                          "K2": d["K2"][0].numpy(), "T_1to2": d["T_1to2"][0].numpy(), "size_A": (args.res, args.res),
                          "size_B": (args.res, args.res)})
        res = pose_benchmark(model, pairs)
        print(json.dumps({"pose_results": res, "acceptance_on_megadepth1500": {k: v[0] for k, v in ACCEPTANCE_POSE.items()},
                          "note": "synthetic planar scenes" + ("" if args.weights else ", RANDOM weights: plumbing check only")}))
        return
    batches = [synthetic_planar_batch(2, args.res, args.res, seed=s) for s in range(args.synthetic)]
    res = benchmark(model, batches)
    print(json.dumps({"results": res, "acceptance_on_megadepth": {k: v[0] for k, v in ACCEPTANCE.items()},
                      "note": "synthetic planar scenes" + ("" if args.weights else ", RANDOM weights: plumbing check only")}))


if __name__ == "__main__":
    main()
