"""Generate tests/golden/* by running the UNMODIFIED reference (/root/reference) on CPU.

Build-container only (needs /root/reference; see tools/ref_import.py).  The committed
fixtures are what travels to the GPU box.  Usage:

    python tools/make_goldens.py contract      # state-dict key/shape contract
    python tools/make_goldens.py ops           # operator-level goldens from reference functions
    python tools/make_goldens.py ops_nearest   # local_correlation(sample_mode="nearest") from the reference's fallback
    python tools/make_goldens.py tiny          # match() 112 -> 168, B=1 symmetric (+ stage tensors)
    python tools/make_goldens.py small         # match() 224 -> 336, B=2, non-symmetric and symmetric coarse-only
    python tools/make_goldens.py full          # match() 560 -> 864, B=1 symmetric (sub-sampled)
    python tools/make_goldens.py odd           # match() 126 x 154 -> 182 x 198: multiples of 14 that are not multiples of 8
    python tools/make_goldens.py mega          # match() 672 -> 1344, B=1 symmetric (tests/test_mega1500.py geometry; sub-sampled)
    python tools/make_goldens.py full8         # match() 560 -> 864, B=8 symmetric, bench.py's rank-0 workload (sub-sampled)
    python tools/make_goldens.py full_coarse   # match() 560 coarse-only, B=1 symmetric (BASELINE config 2 geometry)
    python tools/make_goldens.py full8_indoor  # same geometry, seeds 2 / 3 (BASELINE config 5 "indoor")
    python tools/make_goldens.py kde           # romatch.utils.kde.kde on seeded match-like points
    python tools/make_goldens.py keypoints     # RegressionMatcher.match_keypoints on a seeded warp + keypoints
    python tools/make_goldens.py keypoints_ties  # the same with duplicate keypoints: every tied pair is returned
    python tools/make_goldens.py vis           # RegressionMatcher.visualize_warp on a seeded warp + images
    python tools/make_goldens.py tinyroma      # TinyRoMa.match / forward with the seeded stand-in XFeat backbone
    python tools/make_goldens.py tinyroma_xfeat  # the same with a backbone of the real XFeat architecture; exact_softmax=True
    python tools/make_goldens.py forward       # forward / forward_symmetric / extract_backbone_features (per-scale corresps)
    python tools/make_goldens.py assets        # tests/golden/pair_{A,B}.png: the decoded pixels of assets/sacre_coeur_{A,B}.jpg
    python tools/make_goldens.py match_path    # RegressionMatcher.match(path, path) / match(PIL, PIL): the transform route
    python tools/make_goldens.py tinyroma_path # TinyRoMa.match(path, path) on the asset pair (A and B of different sizes)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_import import install_stubs, build_reference_matcher  # noqa: E402
from roma_amd import synthetic  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def np32(t):
    return t.detach().float().contiguous().numpy()


def contract():
    install_stubs()
    from romatch.models.transformer import vit_large
    from romatch.models.model_zoo import roma_models as rm
    d = vit_large(img_size=518, patch_size=14, init_values=1.0, ffn_layer="mlp", block_chunks=0).state_dict()
    orig = rm.RegressionMatcher.load_state_dict
    rm.RegressionMatcher.load_state_dict = lambda self, w, **k: None
    m = rm.roma_model(resolution=(112, 112), upsample_preds=True, device="cpu", weights=None,
                      dinov2_weights=d, upsample_res=(168, 168), use_custom_corr=False)
    rm.RegressionMatcher.load_state_dict = orig
    out = {"matcher": {k: list(v.shape) for k, v in m.state_dict().items()},
           "dinov2": {k: list(v.shape) for k, v in d.items()}}
    with open(os.path.join(GOLD, "state_dict_contract.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("contract:", len(out["matcher"]), len(out["dinov2"]))


def ops():
    """Operator goldens straight from reference functions/modules (no match())."""
    install_stubs()
    from romatch.utils.local_correlation import local_correlation
    from romatch.utils.utils import cls_to_flow_refine
    g = np.random.Generator(np.random.PCG64(123))

    def rn(*s, std=1.0):
        return torch.from_numpy(g.standard_normal(size=s, dtype=np.float32) * np.float32(std))

    out = {}
    # local correlation: (r, C, h, w) small versions of the three real configurations
    for name, (r, C, h, w) in {"lc_r7": (7, 64, 10, 12), "lc_r3": (3, 128, 14, 14), "lc_r2": (2, 32, 20, 24)}.items():
        B = 2
        f0, f1 = rn(B, C, h, w), rn(B, C, h, w)
        ys = torch.linspace(-1 + 1 / h, 1 - 1 / h, h)
        xs = torch.linspace(-1 + 1 / w, 1 - 1 / w, w)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        grid = torch.stack((gx, gy))[None].expand(B, 2, h, w)
        warp = grid + rn(B, 2, h, w, std=0.35)          # includes out-of-range taps (zero padding)
        warp[0, :, 0, 0] = torch.tensor([-1.7, 0.2])       # far outside
        warp[0, :, 0, 1] = torch.tensor([1.0 - 1.0 / w, -1.0 + 1.0 / h])  # exact pixel centre (integer coords)
        corr = local_correlation(f0, f1, r, warp, use_custom_corr=False)
        out[name + "_f0"], out[name + "_f1"], out[name + "_warp"], out[name + "_corr"] = map(np32, (f0, f1, warp, corr))
    # cls_to_flow_refine
    cls = rn(2, 4096, 6, 5, std=3.0)
    cls[0, 0, 0, 0] = 40.0      # mode at class 0   (clamped neighbours)
    cls[0, 4095, 0, 1] = 40.0   # mode at last class
    cls[0, 63, 0, 2] = 40.0     # row wrap-around neighbour
    out["c2f_cls"], out["c2f_flow"] = np32(cls), np32(cls_to_flow_refine(cls))
    np.savez_compressed(os.path.join(GOLD, "ops_reference.npz"), **out)
    print("ops:", {k: v.shape for k, v in out.items()})


def ops_nearest():
    """sample_mode="nearest" of local_correlation (local_correlation.py:19,30,85) from the reference's torch fallback,
    incl. exact half-pixel ties (nearbyint: to even), exact centres and out-of-range taps."""
    install_stubs()
    from romatch.utils.local_correlation import local_correlation
    g = np.random.Generator(np.random.PCG64(321))

    def rn(*s, std=1.0):
        return torch.from_numpy(g.standard_normal(size=s, dtype=np.float32) * np.float32(std))

    out = {}
    for name, (r, C, h, w) in {"nn_r3": (3, 64, 12, 12), "nn_r2": (2, 32, 16, 20)}.items():
        B = 2
        f0, f1 = rn(B, C, h, w), rn(B, C, h, w)
        ys = torch.linspace(-1 + 1 / h, 1 - 1 / h, h)
        xs = torch.linspace(-1 + 1 / w, 1 - 1 / w, w)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        warp = torch.stack((gx, gy))[None].expand(B, 2, h, w) + rn(B, 2, h, w, std=0.3)
        warp[0, :, 0, 0] = torch.tensor([-0.5, 0.5])        # w = 12: ix = 2.5 exactly (tie -> 2); h = 12: iy = 8.5 (tie -> 8)
        warp[0, :, 0, 1] = torch.tensor([1.0 - 1.0 / w, -1.0 + 1.0 / h])  # exact pixel centre
        warp[0, :, 0, 2] = torch.tensor([-1.9, 0.1])        # far outside
        warp[1, :, 1, 1] = torch.tensor([0.0, 0.0])         # ix = (w - 1) / 2: a tie for even w
        corr = local_correlation(f0, f1, r, warp, use_custom_corr=False, sample_mode="nearest")
        out[name + "_f0"], out[name + "_f1"], out[name + "_warp"], out[name + "_corr"] = map(np32, (f0, f1, warp, corr))
    np.savez_compressed(os.path.join(GOLD, "ops_nearest_reference.npz"), **out)
    print("ops_nearest:", {k: v.shape for k, v in out.items()})


def _run_reference(cfg_name, coarse, up, B, symmetric, upsample_preds, seed_w, seed_in, capture=True):
    sd = synthetic.make_matcher_state_dict(seed_w)
    dsd = synthetic.make_dinov2_state_dict(seed_w)
    hw = lambda v: (v, v) if isinstance(v, int) else tuple(v)  # noqa: E731
    m = build_reference_matcher(sd, dsd, hw(coarse), hw(up), symmetric=symmetric,
                                upsample_preds=upsample_preds)
    inp = synthetic.make_inputs(B, coarse, up if upsample_preds else None, seed=seed_in)
    stages = {}
    if capture:
        calls = {"n": 0}

        def hook_ref(scale):
            def fn(mod, args, out):
                k = calls.get(("cnt", scale), 0)
                calls[("cnt", scale)] = k + 1
                stages[f"ref{scale}_call{k}_dflow"] = np32(out[0])
                stages[f"ref{scale}_call{k}_dcert"] = np32(out[1])
            return fn

        for s in ["16", "8", "4", "2", "1"]:
            m.decoder.conv_refiner[s].register_forward_hook(hook_ref(s))
        def gp_hook(mod, a, o):
            stages["gp16"] = np32(o)
        m.decoder.gps["16"].register_forward_hook(gp_hook)

        def tdec_hook(mod, a, o):
            stages["cls16_argmax"] = o[0].argmax(dim=1).numpy().astype(np.int32)
            top2 = o[0].topk(2, dim=1).values
            stages["cls16_top2gap"] = np32(top2[:, 0] - top2[:, 1])
            stages["gm_cert16"] = np32(o[1])
            from romatch.utils.utils import cls_to_flow_refine
            stages["gm_flow16"] = np32(cls_to_flow_refine(o[0]).permute(0, 3, 1, 2))  # matcher.py:478-481
        m.decoder.embedding_decoder.register_forward_hook(tdec_hook)
        def proj_hook(mod, a, o):
            if "proj16_first" not in stages:
                stages["proj16_first"] = np32(o)
        m.decoder.proj["16"].register_forward_hook(proj_hook)
    t = time.time()
    kw = {}
    if upsample_preds:
        kw = dict(im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
    warp, cert = m.match(inp["im_A"], inp["im_B"], **kw)
    dt = time.time() - t
    print(f"{cfg_name}: reference match() {dt:.1f}s  warp {tuple(warp.shape)} cert {tuple(cert.shape)}")
    return warp, cert, stages, dt


def tiny():
    warp, cert, stages, dt = _run_reference("tiny", 112, 168, 1, True, True, 0, 1)
    np.savez_compressed(os.path.join(GOLD, "match_tiny.npz"), warp=np32(warp), certainty=np32(cert), **stages)
    meta = dict(coarse=112, up=168, B=1, symmetric=True, upsample_preds=True, seed_w=0, seed_in=1,
                min_top2_gap=float(stages["cls16_top2gap"].min()), ref_seconds=dt,
                torch=torch.__version__)
    json.dump(meta, open(os.path.join(GOLD, "match_tiny.json"), "w"), indent=1)
    print(meta)


def small():
    out = {}
    warp, cert, st, _ = _run_reference("small_nonsym", 224, 336, 2, False, True, 3, 4)
    out["nonsym_warp"], out["nonsym_cert"] = np32(warp)[:, ::3, ::3], np32(cert)[:, ::3, ::3]
    out["nonsym_gap"] = st["cls16_top2gap"]
    warp, cert, st, _ = _run_reference("small_coarse_sym", 224, 336, 2, True, False, 3, 4)
    out["coarse_warp"], out["coarse_cert"] = np32(warp)[:, ::3, ::3], np32(cert)[:, ::3, ::3]
    out["coarse_gap"] = st["cls16_top2gap"]
    np.savez_compressed(os.path.join(GOLD, "match_small.npz"), **out)
    meta = dict(coarse=224, up=336, B=2, seed_w=3, seed_in=4, subsample=3,
                min_gap_nonsym=float(out["nonsym_gap"].min()), min_gap_coarse=float(out["coarse_gap"].min()))
    json.dump(meta, open(os.path.join(GOLD, "match_small.json"), "w"), indent=1)
    print(meta)


def full():
    torch.set_num_threads(os.cpu_count())
    warp, cert, st, dt = _run_reference("full", 560, 864, 1, True, True, 0, 1)
    w, c = np32(warp), np32(cert)
    out = dict(warp_sub=w[:, ::8, ::8], cert_sub=c[:, ::8, ::8],
               warp_rowsum=w.sum(axis=(2, 3), dtype=np.float64), cert_rowsum=c.sum(axis=2, dtype=np.float64),
               cls16_argmax=st["cls16_argmax"], cls16_top2gap=st["cls16_top2gap"], gm_cert16=st["gm_cert16"],
               gm_flow16=st["gm_flow16"], gp16_sub=st["gp16"][:, ::8])
    for k, v in st.items():
        if k.startswith("ref") and v.size <= 200000:
            out[k] = v
    np.savez_compressed(os.path.join(GOLD, "match_full.npz"), **out)
    meta = dict(coarse=560, up=864, B=1, symmetric=True, seed_w=0, seed_in=1, subsample=8,
                min_top2_gap=float(st["cls16_top2gap"].min()), ref_seconds=dt,
                threads=torch.get_num_threads(), torch=torch.__version__)
    json.dump(meta, open(os.path.join(GOLD, "match_full.json"), "w"), indent=1)
    print(meta)


def odd():
    """Resolutions that are multiples of 14 but NOT of 8 (roma_models.py:58-59 accepts them; the VGG pyramid then has the
    floor-divided sizes of its max-pools, and every resize in the decoder goes to those sizes): coarse 126 x 154 ->
    upsample 182 x 198 (no multiple-of-anything requirement on the upsample side), B = 1 symmetric, non-square."""
    warp, cert, stages, dt = _run_reference("odd", (126, 154), (182, 198), 1, True, True, 0, 5)
    np.savez_compressed(os.path.join(GOLD, "match_odd.npz"), warp=np32(warp), certainty=np32(cert),
                        cls16_argmax=stages["cls16_argmax"], cls16_top2gap=stages["cls16_top2gap"],
                        gm_flow16=stages["gm_flow16"], gm_cert16=stages["gm_cert16"])
    meta = dict(coarse=[126, 154], up=[182, 198], B=1, symmetric=True, upsample_preds=True, seed_w=0, seed_in=5,
                min_top2_gap=float(stages["cls16_top2gap"].min()), ref_seconds=dt, torch=torch.__version__)
    json.dump(meta, open(os.path.join(GOLD, "match_odd.json"), "w"), indent=1)
    print(meta)


def mega():
    """The geometry of the reference's accuracy tests (tests/test_mega1500.py:12-21: coarse 672, upsample 1344), B = 1
    symmetric fp32 on the synthetic weights: sub-sampled outputs + row checksums + the coarse match."""
    torch.set_num_threads(os.cpu_count())
    warp, cert, st, dt = _run_reference("mega", 672, 1344, 1, True, True, 0, 7)
    w, c = np32(warp), np32(cert)
    out = dict(warp_sub=w[:, ::8, ::8], cert_sub=c[:, ::8, ::8],
               warp_rowsum=w.sum(axis=(2, 3), dtype=np.float64), cert_rowsum=c.sum(axis=2, dtype=np.float64),
               cls16_argmax=st["cls16_argmax"], cls16_top2gap=st["cls16_top2gap"], gm_cert16=st["gm_cert16"],
               gm_flow16=st["gm_flow16"])
    np.savez_compressed(os.path.join(GOLD, "match_mega.npz"), **out)
    meta = dict(coarse=672, up=1344, B=1, symmetric=True, seed_w=0, seed_in=7, subsample=8,
                min_top2_gap=float(st["cls16_top2gap"].min()), ref_seconds=dt,
                threads=torch.get_num_threads(), torch=torch.__version__)
    json.dump(meta, open(os.path.join(GOLD, "match_mega.json"), "w"), indent=1)
    print(meta)


def block_sums(w, c, blk=8):
    """{warp_blocksum [B, H/8, 2W/8, 2], cert_blocksum [B, H/8, 2W/8]} in f64 (symmetric output, H and W multiples of 8)."""
    B, H, W2, _ = w.shape
    W = W2 // 2
    pred = np.concatenate([w[:, :, :W, 2:], w[:, :, W:, :2]], axis=2).astype(np.float64)
    wb = pred.reshape(B, H // blk, blk, W2 // blk, blk, 2).sum(axis=(2, 4))
    cb = c.astype(np.float64).reshape(B, H // blk, blk, W2 // blk, blk).sum(axis=(2, 4))
    return dict(warp_blocksum=wb, cert_blocksum=cb)


def full8(name="match_full8", seed_w=0, seed_in=1):
    """BASELINE configs 3 / 5 geometry: B=8 symmetric 560 -> 864 fp32 (16 directed pairs: batch-index arithmetic
    (b + B) mod 2B at full size).  Default seeds = bench.py's rank-0 workload, so the bench line can carry a parity
    object for the timed configuration; `full8 indoor` (seed 2 / 3) is the BASELINE config-5 "indoor" fixture."""
    torch.set_num_threads(os.cpu_count())
    warp, cert, st, dt = _run_reference(name, 560, 864, 8, True, True, seed_w, seed_in)
    w, c = np32(warp), np32(cert)
    out = dict(warp_sub=w[:, ::8, ::8], cert_sub=c[:, ::8, ::8],
               warp_rowsum=w.sum(axis=(2, 3), dtype=np.float64), cert_rowsum=c.sum(axis=2, dtype=np.float64),
               cls16_argmax=st["cls16_argmax"], cls16_top2gap=st["cls16_top2gap"], gm_cert16=st["gm_cert16"],
               gm_flow16=st["gm_flow16"])
    # round 6: EVERY output pixel of the benchmark geometry enters a gated quantity - f64 sums over 8 x 8 pixel blocks of
    # the certainty and of the two predicted warp channels (left half: [..., 2:], right half: [..., :2]; the other two are
    # the constant grid, compared exactly elsewhere); the 1/8 lattice above holds 1 pixel in 64
    out.update(block_sums(w, c))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    meta = dict(coarse=560, up=864, B=8, symmetric=True, seed_w=seed_w, seed_in=seed_in, subsample=8,
                min_top2_gap=float(st["cls16_top2gap"].min()), ref_seconds=dt,
                threads=torch.get_num_threads(), torch=torch.__version__)
    json.dump(meta, open(os.path.join(GOLD, name + ".json"), "w"), indent=1)
    print(meta)


def full_coarse():
    """BASELINE config 2 geometry: coarse-only (upsample_preds=False) 560 x 560, B=1 symmetric, seeds 0 / 1."""
    torch.set_num_threads(os.cpu_count())
    warp, cert, st, dt = _run_reference("full_coarse", 560, 864, 1, True, False, 0, 1)
    w, c = np32(warp), np32(cert)
    out = dict(warp_sub=w[:, ::8, ::8], cert_sub=c[:, ::8, ::8],
               warp_rowsum=w.sum(axis=(2, 3), dtype=np.float64), cert_rowsum=c.sum(axis=2, dtype=np.float64),
               cls16_argmax=st["cls16_argmax"], cls16_top2gap=st["cls16_top2gap"], gm_cert16=st["gm_cert16"],
               gm_flow16=st["gm_flow16"])
    np.savez_compressed(os.path.join(GOLD, "match_full_coarse.npz"), **out)
    meta = dict(coarse=560, B=1, symmetric=True, upsample_preds=False, seed_w=0, seed_in=1, subsample=8,
                min_top2_gap=float(st["cls16_top2gap"].min()), ref_seconds=dt,
                threads=torch.get_num_threads(), torch=torch.__version__)
    json.dump(meta, open(os.path.join(GOLD, "match_full_coarse.json"), "w"), indent=1)
    print(meta)


def full8_indoor():
    full8("match_full8_indoor", 2, 3)


def kde_golden():
    """Reference romatch.utils.kde.kde on seeded match-like points: f32 (half=False), the reference's default fp16
    evaluation (half=True) and the `down` sub-sampling."""
    install_stubs()
    from romatch.utils.kde import kde as ref_kde
    g = torch.Generator().manual_seed(7)
    n = 1500
    # clustered + spread points in [-1,1]^4, like warp samples: dense blobs give densities well above 10
    centers = torch.rand(12, 4, generator=g) * 1.6 - 0.8
    x = centers[torch.randint(0, 12, (n,), generator=g)] + 0.05 * torch.randn(n, 4, generator=g)
    x[: n // 5] = torch.rand(n // 5, 4, generator=g) * 2 - 1
    out = dict(x=np32(x), density_f32=np32(ref_kde(x, std=0.1, half=False)),
               density_half=np32(ref_kde(x, std=0.1, half=True).float()),
               density_f32_down3=np32(ref_kde(x, std=0.1, half=False, down=3)),
               density_f32_std025=np32(ref_kde(x, std=0.25, half=False)))
    np.savez_compressed(os.path.join(GOLD, "kde_reference.npz"), **out)
    print({k: (v.shape, float(v.min()), float(v.max())) for k, v in out.items()})


def keypoints_golden():
    """Reference RegressionMatcher.match_keypoints on a seeded smooth warp and jittered keypoint sets."""
    install_stubs()
    from romatch.models.matcher import RegressionMatcher
    g = torch.Generator().manual_seed(11)
    H, W = 96, 128
    ys, xs = torch.meshgrid(torch.linspace(-1 + 1 / H, 1 - 1 / H, H), torch.linspace(-1 + 1 / W, 1 - 1 / W, W), indexing="ij")
    bx = 0.9 * xs + 0.08 * torch.sin(3.0 * ys) + 0.03
    by = 0.85 * ys - 0.06 * torch.cos(2.0 * xs) - 0.02
    warp = torch.stack([xs, ys, bx, by], dim=-1)
    cert = torch.sigmoid(4.0 * (1.0 - (xs ** 2 + ys ** 2)))  # confident in the centre
    na, nb = 700, 900
    x_A = torch.rand(na, 2, generator=g) * 1.9 - 0.95
    wA = torch.nn.functional.grid_sample(warp[..., 2:].permute(2, 0, 1)[None], x_A[None, None], align_corners=False)[0, :, 0].mT
    x_B = torch.cat([wA[:500] + 0.002 * torch.randn(500, 2, generator=g), torch.rand(nb - 500, 2, generator=g) * 2 - 1])
    x_B = x_B[torch.randperm(nb, generator=g)]
    out = dict(warp=np32(warp), cert=np32(cert), x_A=np32(x_A), x_B=np32(x_B))
    for name, kw in (("default", {}), ("loose", dict(max_dist=0.02, cert_th=0.6))):
        iA, iB = RegressionMatcher.match_keypoints(None, x_A, x_B, warp, cert, return_tuple=True, return_inds=True, **kw)
        out["inds_A_" + name], out["inds_B_" + name] = iA.numpy().astype(np.int64), iB.numpy().astype(np.int64)
        print(name, len(iA))
    np.savez_compressed(os.path.join(GOLD, "keypoints_reference.npz"), **out)


def keypoints_ties_golden():
    """match_keypoints with DUPLICATE keypoints (a detector that reports one location at several scales): the reference's
    torch.nonzero returns every tied mutual pair (matcher.py:756-762).  Same warp as keypoints_golden; every third
    keypoint of B and every fifth of A is repeated."""
    install_stubs()
    from romatch.models.matcher import RegressionMatcher
    g = torch.Generator().manual_seed(12)
    H, W = 96, 128
    ys, xs = torch.meshgrid(torch.linspace(-1 + 1 / H, 1 - 1 / H, H), torch.linspace(-1 + 1 / W, 1 - 1 / W, W), indexing="ij")
    bx = 0.9 * xs + 0.08 * torch.sin(3.0 * ys) + 0.03
    by = 0.85 * ys - 0.06 * torch.cos(2.0 * xs) - 0.02
    warp = torch.stack([xs, ys, bx, by], dim=-1)
    cert = torch.sigmoid(4.0 * (1.0 - (xs ** 2 + ys ** 2)))
    x_A = torch.rand(300, 2, generator=g) * 1.9 - 0.95
    wA = torch.nn.functional.grid_sample(warp[..., 2:].permute(2, 0, 1)[None], x_A[None, None], align_corners=False)[0, :, 0].mT
    x_B = torch.cat([wA[:200] + 0.002 * torch.randn(200, 2, generator=g), torch.rand(150, 2, generator=g) * 2 - 1])
    x_B = torch.cat([x_B, x_B[::3], x_B[::7]])           # duplicates (some points three times)
    x_B = x_B[torch.randperm(len(x_B), generator=g)]
    x_A = torch.cat([x_A, x_A[::5]])
    x_A = x_A[torch.randperm(len(x_A), generator=g)]
    out = dict(warp=np32(warp), cert=np32(cert), x_A=np32(x_A), x_B=np32(x_B))
    for name, kw in (("default", {}), ("loose", dict(max_dist=0.02, cert_th=0.6))):
        iA, iB = RegressionMatcher.match_keypoints(None, x_A, x_B, warp, cert, return_tuple=True, return_inds=True, **kw)
        out["inds_A_" + name], out["inds_B_" + name] = iA.numpy().astype(np.int64), iB.numpy().astype(np.int64)
        print(name, len(iA), "pairs,", len(torch.unique(iA)), "distinct A keypoints")
    np.savez_compressed(os.path.join(GOLD, "keypoints_ties_reference.npz"), **out)


def tinyroma_golden():
    """The reference's own TinyRoMa (romatch/models/tiny.py; tiny_roma_v1_model with exact_softmax=False, eval) on seeded
    image pairs, with roma_amd.synthetic.XFeatStandIn in place of the un-vendored XFeat hub model and seeded matcher
    weights.  Stored: inputs, the backbone features (so the device test can start from identical features), both
    correspondence levels and match() outputs, for two 96 x 128 pairs and a 100 x 150 pair (pre-processing resize).
    Every pair is run on its own (B = 1): tiny.py:137 broadcasts the batch axis of the arg-max probability against the
    channel axis of the grid, so the reference's batched result is only well defined for B = 1 (see oracle/tiny_oracle.py)."""
    install_stubs()
    import types
    if "torchvision.transforms" not in sys.modules:  # tiny.py imports ToTensor at module level
        tv = sys.modules.get("torchvision") or types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        tr.ToTensor = lambda: (lambda im: torch.from_numpy(np.array(im)).permute(2, 0, 1).float() / 255)
        tv.transforms = tr
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tr
    from romatch.models.tiny import TinyRoMa
    from roma_amd import synthetic
    sd = synthetic.make_tiny_state_dict(0)
    model = TinyRoMa(xfeat=synthetic.XFeatStandIn(0), freeze_xfeat=True, exact_softmax=False)
    missing = model.load_state_dict(sd, strict=True)
    model.train(False)
    out = {}
    for tag, (b, h, w) in (("a", (2, 96, 128)), ("b", (1, 100, 150))):
        inp = synthetic.make_tiny_inputs(b, h, w, seed=3 if tag == "a" else 4)
        per = {k: [] for k in ("fineA", "fineB", "coarseA", "coarseB", "flow8", "cert8", "flow4", "cert4", "warp", "cert")}
        for p in range(b):
            ia, ib = inp["im_A"][p:p + 1], inp["im_B"][p:p + 1]
            with torch.inference_mode():
                x = torch.cat([model.preprocess_tensor(ia)[0], model.preprocess_tensor(ib)[0]], dim=0)
                fine, coarse = model.forward_single(x)
                corr = model.forward({"im_A": ia, "im_B": ib})
                warp, cert = model.match(ia, ib, batched=True)
            for k, v in (("fineA", fine[:1]), ("fineB", fine[1:]), ("coarseA", coarse[:1]), ("coarseB", coarse[1:]),
                         ("flow8", corr[8]["flow"]), ("cert8", corr[8]["certainty"]), ("flow4", corr[4]["flow"]),
                         ("cert4", corr[4]["certainty"]), ("warp", warp), ("cert", cert)):
                per[k].append(v.clone())
        cat = {k: torch.cat(v, dim=0) for k, v in per.items()}
        out.update({f"{tag}_im_A": np32(inp["im_A"]), f"{tag}_im_B": np32(inp["im_B"]),
                    f"{tag}_feat_fine": np32(torch.cat((cat["fineA"], cat["fineB"]), dim=0)),          # [A pairs ..., B pairs ...]
                    f"{tag}_feat_coarse": np32(torch.cat((cat["coarseA"], cat["coarseB"]), dim=0)),
                    f"{tag}_flow8": np32(cat["flow8"]), f"{tag}_cert8": np32(cat["cert8"]), f"{tag}_flow4": np32(cat["flow4"]),
                    f"{tag}_cert4": np32(cat["cert4"]), f"{tag}_warp": np32(cat["warp"]), f"{tag}_cert": np32(cat["cert"])})
    # a demo-sized pair (BASELINE config 1 runs 'assets/sacre_coeur_A/B', ~ 480 x 640): end to end only, outputs 1/4 sub-sampled
    inp = synthetic.make_tiny_inputs(1, 480, 640, seed=5)
    with torch.inference_mode():
        warp, cert = model.match(inp["im_A"], inp["im_B"], batched=True)
    out.update({"c_seed": np.array([5]), "c_warp_sub": np32(warp[:, ::4, ::4]), "c_cert_sub": np32(cert[:, ::4, ::4])})
    np.savez_compressed(os.path.join(GOLD, "tiny_reference.npz"), **out)
    print("tiny_reference.npz", {k: v.shape for k, v in out.items()}, missing)


def tinyroma_xfeat_golden():
    """The reference's TinyRoMa with a backbone of the real XFeat ARCHITECTURE (roma_amd.synthetic.XFeatArch: BasicLayer
    stacks with BatchNorm, stride-2 and 1 x 1 layers; seeded weights), for the device replay of forward_single
    (tiny.py:81-99) and for the exact_softmax=True branch of pos_embed (tiny.py:139-141).  Per case: the backbone features
    of both images, both correspondence levels and match()."""
    install_stubs()
    import types
    if "torchvision.transforms" not in sys.modules:
        tv = sys.modules.get("torchvision") or types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        tr.ToTensor = lambda: (lambda im: torch.from_numpy(np.array(im)).permute(2, 0, 1).float() / 255)
        tv.transforms = tr
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tr
    from romatch.models.tiny import TinyRoMa
    from roma_amd import synthetic
    sd = synthetic.make_tiny_state_dict(0)
    out = {}
    for tag, (h, w, exact, seed) in {"x": (96, 128, False, 6), "e": (96, 128, True, 6), "y": (100, 150, False, 7)}.items():
        model = TinyRoMa(xfeat=synthetic.XFeatArch(0), freeze_xfeat=True, exact_softmax=exact)
        model.load_state_dict(sd, strict=True)
        model.train(False)
        inp = synthetic.make_tiny_inputs(1, h, w, seed=seed)
        ia, ib = inp["im_A"], inp["im_B"]
        with torch.inference_mode():
            x = torch.cat([model.preprocess_tensor(ia)[0], model.preprocess_tensor(ib)[0]], dim=0)
            fine, coarse = model.forward_single(x)
            corr = model.forward({"im_A": ia, "im_B": ib})
            warp, cert = model.match(ia, ib, batched=True)
        out.update({f"{tag}_im_A": np32(ia), f"{tag}_im_B": np32(ib), f"{tag}_feat_fine": np32(fine), f"{tag}_feat_coarse": np32(coarse),
                    f"{tag}_flow8": np32(corr[8]["flow"]), f"{tag}_cert8": np32(corr[8]["certainty"]),
                    f"{tag}_flow4": np32(corr[4]["flow"]), f"{tag}_cert4": np32(corr[4]["certainty"]),
                    f"{tag}_warp": np32(warp), f"{tag}_cert": np32(cert)})
    np.savez_compressed(os.path.join(GOLD, "tiny_xfeat_reference.npz"), **out)
    print("tiny_xfeat_reference.npz", {k: v.shape for k, v in out.items()})


def forward_golden():
    """RegressionMatcher.forward / forward_symmetric / extract_backbone_features of the unmodified reference
    (matcher.py:585-596, 631-670), 112 -> 168, seeded weights: the per-scale `corresps` dict of the coarse pass (symmetric,
    the scale factor match() uses), of the upsample pass seeded with the coarse pass's finest correspondences (matcher.py:870-889),
    of the non-symmetric forward at B = 2 with forward()'s default scale_factor = 1, and the feature pyramid."""
    import math
    sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)
    m = build_reference_matcher(sd, dsd, (112, 112), (168, 168), symmetric=True, upsample_preds=True)
    out = {}
    inp = synthetic.make_inputs(1, 112, 168, seed=1)
    with torch.inference_mode():
        sf = math.sqrt(112 * 112 / 560 ** 2)
        cor = m.forward_symmetric({"im_A": inp["im_A"], "im_B": inp["im_B"]}, scale_factor=sf)
        for s_, v in cor.items():
            assert set(v.keys()) == {"certainty", "flow"}, v.keys()  # eval mode: nothing else (matcher.py:469-512)
            out[f"sym_flow{s_}"], out[f"sym_cert{s_}"] = np32(v["flow"]), np32(v["certainty"])
        sf2 = math.sqrt(168 * 168 / 560 ** 2)
        cu = m.forward_symmetric({"im_A": inp["im_A_high_res"], "im_B": inp["im_B_high_res"], "corresps": cor[1]},
                                 upsample=True, batched=True, scale_factor=sf2)
        assert sorted(cu.keys()) == [1, 2, 4, 8]
        for s_, v in cu.items():
            out[f"up_flow{s_}"], out[f"up_cert{s_}"] = np32(v["flow"]), np32(v["certainty"])
        inp2 = synthetic.make_inputs(2, 112, None, seed=7)
        cf = m.forward({"im_A": inp2["im_A"], "im_B": inp2["im_B"]})  # batched=True, scale_factor=1 defaults
        for s_, v in cf.items():
            out[f"fwd_flow{s_}"], out[f"fwd_cert{s_}"] = np32(v["flow"]), np32(v["certainty"])
        fp = m.extract_backbone_features({"im_A": inp["im_A"], "im_B": inp["im_B"]})
        sub = {16: 1, 8: 1, 4: 2, 2: 2, 1: 4}
        for s_, f in fp.items():
            out[f"feat{s_}"] = np32(f[:, :, ::sub[s_], ::sub[s_]])
            out[f"feat{s_}_shape"] = np.array(f.shape)
        fu = m.extract_backbone_features({"im_A": inp["im_A_high_res"], "im_B": inp["im_B_high_res"]}, upsample=True)
        out["feat_up_scales"] = np.array(sorted(fu.keys()))
        out["feat_up8"] = np32(fu[8])
    np.savez_compressed(os.path.join(GOLD, "match_forward.npz"), **out)
    print("match_forward.npz", {k: v.shape for k, v in out.items()})


ASSET_PAIR = ("sacre_coeur_A.jpg", "sacre_coeur_B.jpg")  # BASELINE config 1's pair: 640 x 480 and 618 x 640 (W x H), RGB


def assets_golden():
    """tests/golden/pair_A.png / pair_B.png: the decoded RGB pixels of the reference's demo pair (assets/sacre_coeur_*.jpg),
    stored lossless so that the GPU box - which has no /root/reference - opens bit-identical pixels through the same
    `Image.open(path).convert("RGB")` the reference's path route starts with (matcher.py:530-547)."""
    from PIL import Image
    for src, dst in zip(ASSET_PAIR, ("pair_A.png", "pair_B.png")):
        im = Image.open(os.path.join("/root/reference/assets", src)).convert("RGB")
        im.save(os.path.join(GOLD, dst), optimize=True)
        back = np.array(Image.open(os.path.join(GOLD, dst)).convert("RGB"))
        assert np.array_equal(back, np.array(im)), "PNG round trip must be lossless"
        print(dst, im.size, os.path.getsize(os.path.join(GOLD, dst)) // 1024, "KiB")


def match_path_golden():
    """The reference's OWN match(path, path) (matcher.py:806-816 coarse transform, 853-868 upsample transform, B = 1,
    _check_input) and match(PIL, PIL) on the demo pair, seeded weights, 112 -> 168 (cheap on CPU; the transform route is
    resolution independent), plus a non-square configuration 112 x 140 -> 168 x 196 and a coarse-only run.  Stored: the
    normalised tensors the reference's transforms produced (get_tuple_transform_ops, utils/utils.py:164-173) - the product's
    _pil_to_normalised is held to them bit for bit on CPU - and warp / certainty."""
    install_stubs()
    from PIL import Image
    from romatch.utils.utils import get_tuple_transform_ops
    pa, pb = (os.path.join(GOLD, n) for n in ("pair_A.png", "pair_B.png"))
    sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)
    out = {}
    for tag, coarse, up in (("sq", (112, 112), (168, 168)), ("rect", (112, 140), (168, 196))):
        m = build_reference_matcher(sd, dsd, coarse, up, symmetric=True, upsample_preds=True)
        ims = (Image.open(pa).convert("RGB"), Image.open(pb).convert("RGB"))
        for nm, res in (("coarse", coarse), ("up", up)):
            ta, tb = get_tuple_transform_ops(resize=res, normalize=True)(ims)
            out[f"{tag}_{nm}_A"], out[f"{tag}_{nm}_B"] = np32(ta), np32(tb)
        t = time.time()
        warp, cert = m.match(pa, pb)
        print(f"match_path {tag}: {time.time() - t:.1f}s", tuple(warp.shape), tuple(cert.shape))
        warp2, cert2 = m.match(*ims)  # the PIL route: same transforms, must be the same result
        assert torch.equal(warp, warp2) and torch.equal(cert, cert2)
        out[f"{tag}_warp"], out[f"{tag}_cert"] = np32(warp), np32(cert)
        if tag == "sq":
            m.upsample_preds = False  # mutable attribute (README.md:82-90): coarse-only from paths
            warp, cert = m.match(pa, pb)
            out["sq_coarse_only_warp"], out["sq_coarse_only_cert"] = np32(warp), np32(cert)
    np.savez_compressed(os.path.join(GOLD, "match_path.npz"), **out)
    print("match_path.npz", {k: v.shape for k, v in out.items()})


def tinyroma_path_golden():
    """BASELINE config 1 as the reference runs it: TinyRoMa.match(path, path) -> match_from_path (tiny.py:193-198: ToTensor,
    no resize, no normalisation) on the demo pair.  A is 480 x 640, B is 640 x 618 -> 640 x 608 after preprocess_tensor, so
    forward() takes the separate forward_single branch (tiny.py:288-290).  Backbone: the real XFeat layer list with seeded
    weights (roma_amd.synthetic.XFeatArch; the hub checkpoint is unobtainable offline).  Outputs 1/4 sub-sampled."""
    install_stubs()
    from romatch.models.tiny import TinyRoMa
    pa, pb = (os.path.join(GOLD, n) for n in ("pair_A.png", "pair_B.png"))
    sd = synthetic.make_tiny_state_dict(0)
    out = {}
    for tag, exact in (("p", False), ("pe", True)):
        model = TinyRoMa(xfeat=synthetic.XFeatArch(0), freeze_xfeat=True, exact_softmax=exact)
        model.load_state_dict(sd, strict=True)
        model.train(False)
        with torch.inference_mode():
            warp, cert = model.match(pa, pb)  # str -> match_from_path -> batched=False
        assert warp.dim() == 3 and cert.dim() == 2
        out[f"{tag}_shape"] = np.array(warp.shape)
        out[f"{tag}_warp_sub"], out[f"{tag}_cert_sub"] = np32(warp[::4, ::4]), np32(cert[::4, ::4])
        out[f"{tag}_warp_rowsum"] = np32(warp.double().sum(dim=(1, 2)).float())
        out[f"{tag}_cert_rowsum"] = np32(cert.double().sum(dim=1).float())
    np.savez_compressed(os.path.join(GOLD, "tiny_path_reference.npz"), **out)
    print("tiny_path_reference.npz", {k: v.shape for k, v in out.items()})


def vis_golden():
    """Reference RegressionMatcher.visualize_warp (matcher.py:936-986) on a seeded smooth symmetric warp, tensor images
    of the warp's resolution, and the non-symmetric form with images of a different resolution."""
    install_stubs()
    from romatch.models.matcher import RegressionMatcher
    g = torch.Generator().manual_seed(23)
    H, W = 48, 64
    ys, xs = torch.meshgrid(torch.linspace(-1 + 1 / H, 1 - 1 / H, H), torch.linspace(-1 + 1 / W, 1 - 1 / W, W), indexing="ij")
    grid = torch.stack((xs, ys), dim=-1)
    tgt_ab = (grid * 0.9 + 0.15 * torch.sin(3 * grid.flip(-1)) + 0.12).float()   # leaves the image at one border (zeros padding)
    tgt_ba = (grid * 1.05 - 0.1 * torch.cos(2 * grid)).float()
    warp = torch.cat((torch.cat((grid, tgt_ab), dim=-1), torch.cat((tgt_ba, grid), dim=-1)), dim=1)  # [H, 2W, 4]
    cert = torch.rand(H, 2 * W, generator=g)
    im_A, im_B = torch.rand(3, H, W, generator=g), torch.rand(3, H, W, generator=g)
    out = {"warp": warp.numpy(), "cert": cert.numpy(), "im_A": im_A.numpy(), "im_B": im_B.numpy()}
    out["vis_sym"] = RegressionMatcher.visualize_warp(None, warp, cert, im_A=im_A, im_B=im_B, device="cpu", symmetric=True).numpy()
    im_B2 = torch.rand(3, 30, 44, generator=g)
    out["im_B2"] = im_B2.numpy()
    out["vis_one"] = RegressionMatcher.visualize_warp(None, warp[:, :W], cert[:, :W], im_A=im_A, im_B=im_B2, device="cpu",
                                                      symmetric=False).numpy()
    np.savez_compressed(os.path.join(GOLD, "visualize_reference.npz"), **out)
    print("visualize_reference.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    for what in sys.argv[1:]:
        {"contract": contract, "ops": ops, "ops_nearest": ops_nearest, "tiny": tiny, "small": small, "full": full, "odd": odd, "mega": mega, "full8": full8, "full8_indoor": full8_indoor, "full_coarse": full_coarse, "kde": kde_golden, "keypoints": keypoints_golden, "keypoints_ties": keypoints_ties_golden, "vis": vis_golden, "tinyroma": tinyroma_golden, "tinyroma_xfeat": tinyroma_xfeat_golden, "forward": forward_golden, "assets": assets_golden, "match_path": match_path_golden, "tinyroma_path": tinyroma_path_golden}[what]()
