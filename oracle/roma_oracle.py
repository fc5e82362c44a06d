"""CPU ORACLE for the RoMa `RegressionMatcher.match()` hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a functional fp32 restatement (torch CPU ops, no nn.Module, no autocast) of the
reference algorithm, each function citing the reference file:line it follows (paths relative
to /root/reference/).  It is the *checker* for the HIP path; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The product
(`roma_amd/`) never imports it and fails loudly when the HIP library is missing.

Parity pinning: the reference's own tests hold no golden vectors for this path (SURVEY.md
section 8c).  The oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, run in the build
container by `tools/make_goldens.py` (stub-import of /root/reference, seeded synthetic
weights) and committed under `tests/golden/`; `tests/test_cpu_oracle.py` checks the oracle
against them (per-stage tensors and final warp/certainty; also `kde`, `sample`, `match_keypoints`, the
post-processing steps restated at the end of this file).  Third-party arithmetic that is
absent from /root/reference: `fused-local-corr` 0.2.2 (CUDA wheel, uv.lock:541-554) - its
semantics are taken from the in-repo torch fallback (romatch/utils/local_correlation.py:39-74),
the reference holds no test comparing the two, so parity at that operator boundary is pinned
to the fallback only.

The CPU reference path is fp32 end to end (roma_models.py:55-56, utils/utils.py:639-653).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

VGG_CONV_IDX = [0, 3, 7, 10, 14, 17, 20, 23, 27, 30, 33, 36]
VGG_POOL_AFTER = {3: 1, 10: 2, 23: 4, 36: 8}  # conv index whose (BN,ReLU) output is emitted at this stride
REFINER_RADIUS = {"16": 7, "8": 3, "4": 2, "2": 0, "1": 0}


# ----------------------------------------------------------------------------- encoder
def _bn_eval(x, sd, prefix, eps=1e-5):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, eps)


def vgg19_pyramid(x, sd) -> Dict[int, torch.Tensor]:
    """romatch/models/encoders.py:17-27 on torchvision vgg19_bn().features[:40]:
    the activation entering each MaxPool is emitted at stride 1,2,4,8."""
    feats = {}
    for idx in VGG_CONV_IDX:
        p = f"encoder.cnn.layers.{idx}"
        x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        x = F.relu(_bn_eval(x, sd, f"encoder.cnn.layers.{idx + 1}"))
        if idx in VGG_POOL_AFTER:
            feats[VGG_POOL_AFTER[idx]] = x
            x = F.max_pool2d(x, 2, 2)
    return feats


def dinov2_interpolate_pos_encoding(pos_embed, h_img, w_img, patch=14):
    """romatch/models/transformer/dinov2.py:166-190 (note: called as (x, w, h) with
    B,nc,w,h = x.shape, so its `w` is the image HEIGHT; bicubic with the +0.1 scale-factor quirk)."""
    N = pos_embed.shape[1] - 1
    npatch = (h_img // patch) * (w_img // patch)
    if npatch == N and h_img == w_img:
        return pos_embed
    pos_embed = pos_embed.float()
    class_pos = pos_embed[:, 0]
    patch_pos = pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    w0 = h_img // patch + 0.1
    h0 = w_img // patch + 0.1
    M = int(math.sqrt(N))
    patch_pos = F.interpolate(patch_pos.reshape(1, M, M, dim).permute(0, 3, 1, 2),
                              scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((class_pos.unsqueeze(0), patch_pos), dim=1)


def _attention(x, qkv_w, qkv_b, proj_w, proj_b, heads):
    """romatch/models/transformer/layers/attention.py:50-63."""
    B, N, C = x.shape
    qkv = F.linear(x, qkv_w, qkv_b).reshape(B, N, 3, heads, C // heads)
    q, k, v = torch.unbind(qkv, 2)
    q, k, v = [t.transpose(1, 2) for t in (q, k, v)]
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, N, C)
    return F.linear(o, proj_w, proj_b)


def _vit_block(x, sd, p, heads, eps, layerscale):
    """romatch/models/transformer/layers/block.py:82-107 (eval branch), mlp.py:35-41,
    layer_scale.py:27-28."""
    h = F.layer_norm(x, (x.shape[-1],), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps)
    h = _attention(h, sd[p + ".attn.qkv.weight"], sd.get(p + ".attn.qkv.bias"),
                   sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"], heads)
    if layerscale:
        h = h * sd[p + ".ls1.gamma"]
    x = x + h
    h = F.layer_norm(x, (x.shape[-1],), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)
    h = F.linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])
    h = F.gelu(h)
    h = F.linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    if layerscale:
        h = h * sd[p + ".ls2.gamma"]
    return x + h


def dinov2_patch_tokens(x, dsd, return_blocks=False):
    """forward_features -> x_norm_patchtokens  (dinov2.py:192-237; patch_embed.py:69-82),
    reshaped to [B,1024,H/14,W/14] as encoders.py:64-65 does."""
    B, _, H, W = x.shape
    t = F.conv2d(x, dsd["patch_embed.proj.weight"], dsd["patch_embed.proj.bias"], stride=14)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((dsd["cls_token"].expand(B, -1, -1), t), dim=1)
    t = t + dinov2_interpolate_pos_encoding(dsd["pos_embed"], H, W)
    blocks = []
    for i in range(24):
        t = _vit_block(t, dsd, f"blocks.{i}", 16, 1e-6, True)
        if return_blocks:
            blocks.append(t)
    t = F.layer_norm(t, (1024,), dsd["norm.weight"], dsd["norm.bias"], 1e-6)
    f16 = t[:, 1:].permute(0, 2, 1).reshape(B, 1024, H // 14, W // 14)
    return (f16, blocks) if return_blocks else f16


def encoder_pyramid(x, sd, dsd, upsample=False):
    """CNNandDinov2.forward, romatch/models/encoders.py:56-68."""
    pyr = vgg19_pyramid(x, sd)
    if not upsample:
        pyr[16] = dinov2_patch_tokens(x, dsd)
    return pyr


# ----------------------------------------------------------------------------- decoder parts
def pixel_grid(b, h, w):
    """(x, y) pixel-centre grid, [b,2,h,w]; romatch/models/matcher.py:136-144, 365-377."""
    ys = torch.linspace(-1 + 1 / h, 1 - 1 / h, h)
    xs = torch.linspace(-1 + 1 / w, 1 - 1 / w, w)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack((gx, gy))[None].expand(b, 2, h, w)


def proj_head(f, sd, scale):
    """Conv1x1 + BatchNorm(eval); roma_models.py:156-160, applied at matcher.py:441-450."""
    p = f"decoder.proj.{scale}"
    return _bn_eval(F.conv2d(f.float(), sd[p + ".0.weight"], sd[p + ".0.bias"]), sd, p + ".1")


def cos_kernel(x, y, T=0.2, eps=1e-6):
    """CosKernel.__call__, matcher.py:191-200."""
    c = torch.einsum("bnd,bmd->bnm", x, y) / (x.norm(dim=-1)[..., None] * y.norm(dim=-1)[:, None] + eps)
    return ((c - 1.0) / T).exp()


def gp_posterior(x, y, sd, sigma_noise=0.1, return_parts=False):
    """GP.forward eval branch, matcher.py:291-323 (+ get_pos_enc 274-289, project_to_basis 264-272)."""
    b, c, h1, w1 = x.shape
    _, _, h2, w2 = y.shape
    coords = pixel_grid(b, h2, w2)
    f = torch.cos(8 * math.pi * F.conv2d(coords, sd["decoder.gps.16.pos_conv.weight"],
                                         sd["decoder.gps.16.pos_conv.bias"]))
    d = f.shape[1]
    xr = x.float().flatten(2).transpose(1, 2)
    yr = y.float().flatten(2).transpose(1, 2)
    fr = f.flatten(2).transpose(1, 2)
    K_yy = cos_kernel(yr, yr)
    K_xy = cos_kernel(xr, yr)
    A = K_yy + sigma_noise * torch.eye(h2 * w2)[None]
    L = torch.linalg.cholesky(A)
    alpha = torch.cholesky_solve(fr.reshape(b, h2 * w2, d), L, upper=False)
    mu = K_xy @ alpha
    out = mu.transpose(1, 2).reshape(b, d, h1, w1)
    if return_parts:
        return out, dict(K_yy=K_yy, K_xy=K_xy, L=L, alpha=alpha, f=fr)
    return out


def transformer_decoder(gp_post, feats, sd):
    """TransformerDecoder.forward, romatch/models/transformer/__init__.py:30-46 with the
    blocks of roma_models.py:75-84 (5 x Block(1024, 8 heads), LN eps 1e-5, no LayerScale)."""
    x = torch.cat((gp_post, feats), dim=1)
    B, C, H, W = x.shape
    t = x.reshape(B, C, H * W).permute(0, 2, 1)
    for i in range(5):
        t = _vit_block(t, sd, f"decoder.embedding_decoder.blocks.{i}", 8, 1e-5, False)
    out = F.linear(t, sd["decoder.embedding_decoder.to_out.weight"], sd["decoder.embedding_decoder.to_out.bias"])
    out = out.permute(0, 2, 1).reshape(B, -1, H, W)
    return out[:, :-1], out[:, -1:]


def cls_to_flow_refine(cls):
    """romatch/utils/utils.py:300-322."""
    B, C, H, W = cls.shape
    res = round(math.sqrt(C))
    lin = torch.linspace(-1 + 1 / res, 1 - 1 / res, steps=res)
    G = torch.meshgrid(lin, lin, indexing="ij")
    G = torch.stack([G[1], G[0]], dim=-1).reshape(C, 2)
    cls = cls.softmax(dim=1)
    mode = cls.max(dim=1).indices
    index = torch.stack((mode - 1, mode, mode + 1, mode - res, mode + res), dim=1).clamp(0, C - 1).long()
    nb = torch.gather(cls, dim=1, index=index)[..., None]
    flow = sum(nb[:, i] * G[index[:, i]] for i in range(5))
    return flow / nb.sum(dim=1)


def local_correlation(f0, f1, r, warp, sample_mode="bilinear"):
    """romatch/utils/local_correlation.py:77-143 with the torch fallback :39-74
    (the semantics the external fused kernel must reproduce).  warp: [B,2,H,W]; sample_mode "bilinear" | "nearest"
    (:19,30,85 - the matcher only ever passes "bilinear", matcher.py:43)."""
    K = (2 * r + 1) ** 2
    B, c, h, w = f0.shape
    warp = warp.permute(0, 2, 3, 1)
    lw = torch.meshgrid(torch.linspace(-2 * r / h, 2 * r / h, 2 * r + 1),
                        torch.linspace(-2 * r / w, 2 * r / w, 2 * r + 1), indexing="ij")
    lw = torch.stack((lw[1], lw[0]), dim=-1).reshape(1, K, 2)
    corr = torch.empty((B, K, h, w), dtype=f0.dtype)
    for i in range(B):
        coords = (warp[i, :, :, None] + lw[:, None, None]).reshape(1, h, w * K, 2)
        wf = F.grid_sample(f1[i:i + 1], coords, padding_mode="zeros", align_corners=False, mode=sample_mode)
        wf = wf.reshape(c, h, w, K)
        corr[i] = (f0[i, ..., None] / (c ** 0.5) * wf).sum(dim=0).permute(2, 0, 1)
    return corr


def _refiner_block(d, sd, p):
    """create_block, matcher.py:92-122: dw5x5(bias) -> BN(eval) -> ReLU -> 1x1(bias)."""
    C = d.shape[1]
    d = F.conv2d(d, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=2, groups=C)
    d = F.relu(_bn_eval(d, sd, p + ".1"))
    return F.conv2d(d, sd[p + ".3.weight"], sd[p + ".3.bias"])


def conv_refiner(x, y, warp, sd, scale, scale_factor, return_parts=False):
    """ConvRefiner.forward, matcher.py:124-179."""
    p = f"decoder.conv_refiner.{scale}"
    b, c, hs, ws = x.shape
    x_hat = F.grid_sample(y, warp.permute(0, 2, 3, 1), align_corners=False, mode="bilinear")
    disp = warp - pixel_grid(b, hs, ws)
    emb = F.conv2d(40 / 32 * scale_factor * disp, sd[p + ".disp_emb.weight"], sd[p + ".disp_emb.bias"])
    r = REFINER_RADIUS[scale]
    if r:
        corr = local_correlation(x, y, r, warp)
        d = torch.cat((x, x_hat, emb, corr), dim=1)
    else:
        corr = None
        d = torch.cat((x, x_hat, emb), dim=1)
    d_in = d
    d = _refiner_block(d, sd, p + ".block1")
    for hb in range(8):
        d = _refiner_block(d, sd, p + f".hidden_blocks.{hb}")
    d = F.conv2d(d.float(), sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"])
    if return_parts:
        return d[:, :-1], d[:, -1:], dict(x_hat=x_hat, emb=emb, corr=corr, d_in=d_in)
    return d[:, :-1], d[:, -1:]


def decoder_forward(f1, f2, sd, upsample=False, flow=None, certainty=None, scale_factor=1.0,
                    stages: Optional[dict] = None):
    """Decoder.forward, matcher.py:395-527 (eval; scales 16,8,4,2,1 or 8,4,2,1 when upsampling)."""
    all_scales = ["16", "8", "4", "2", "1"] if not upsample else ["8", "4", "2", "1"]
    sizes = {s: f1[s].shape[-2:] for s in f1}
    h, w = sizes[1]
    b = f1[1].shape[0]
    coarsest = int(all_scales[0])
    corresps = {}
    if not upsample:
        flow = pixel_grid(b, *sizes[coarsest])
        certainty = 0.0
    else:
        flow = F.interpolate(flow, size=sizes[coarsest], align_corners=False, mode="bilinear")
        certainty = F.interpolate(certainty, size=sizes[coarsest], align_corners=False, mode="bilinear")
    for new_scale in all_scales:
        ins = int(new_scale)
        f1_s, f2_s = proj_head(f1[ins], sd, new_scale), proj_head(f2[ins], sd, new_scale)
        if ins == 16:
            gp = gp_posterior(f1_s, f2_s, sd)
            cls, certainty = transformer_decoder(gp, f1_s, sd)
            flow = cls_to_flow_refine(cls).permute(0, 3, 1, 2)
            if stages is not None:
                stages.update(gp16=gp, cls16=cls, gm_cert16=certainty, gm_flow16=flow, proj16=f1_s)
        dflow, dcert = conv_refiner(f1_s, f2_s, flow, sd, new_scale, scale_factor)
        disp = ins * torch.stack((dflow[:, 0].float() / (4 * w), dflow[:, 1].float() / (4 * h)), dim=1)
        flow = flow + disp
        certainty = certainty + dcert
        corresps[ins] = {"certainty": certainty, "flow": flow}
        if new_scale != "1":
            flow = F.interpolate(flow, size=sizes[ins // 2], mode="bilinear")
            certainty = F.interpolate(certainty, size=sizes[ins // 2], mode="bilinear")
    return corresps


# ----------------------------------------------------------------------------- match()
def _forward(im_A, im_B, sd, dsd, symmetric, upsample, scale_factor, corresps=None, stages=None):
    """forward / forward_symmetric, matcher.py:631-670 (batched)."""
    X = torch.cat((im_A, im_B), dim=0)
    pyr = encoder_pyramid(X, sd, dsd, upsample=upsample)
    if stages is not None and not upsample:
        stages.update({f"feat{s}": v for s, v in pyr.items()})
    if symmetric:
        f_q = pyr
        f_s = {s: torch.cat((v.chunk(2)[1], v.chunk(2)[0]), dim=0) for s, v in pyr.items()}
    else:
        f_q = {s: v.chunk(2)[0] for s, v in pyr.items()}
        f_s = {s: v.chunk(2)[1] for s, v in pyr.items()}
    kw = dict(flow=corresps["flow"], certainty=corresps["certainty"]) if corresps is not None else {}
    return decoder_forward(f_q, f_s, sd, upsample=upsample, scale_factor=scale_factor,
                           stages=stages if not upsample else None, **kw)


@torch.inference_mode()
def match(im_A, im_B, sd, dsd, im_A_high_res=None, im_B_high_res=None, symmetric=True,
          upsample_preds=True, attenuate_cert=True, upsample_res=None, stages: Optional[dict] = None):
    """RegressionMatcher.match for tensor inputs, matcher.py:779-934.  Returns (warp, certainty)."""
    b, _, hs, ws = im_A.shape
    scale_factor = math.sqrt(hs * ws / (560 ** 2))
    corresps = _forward(im_A, im_B, sd, dsd, symmetric, False, scale_factor, stages=stages)
    if stages is not None:
        for s in corresps:
            stages[f"p1_flow{s}"] = corresps[s]["flow"]
            stages[f"p1_cert{s}"] = corresps[s]["certainty"]
    if upsample_preds:
        assert im_A_high_res is not None and im_B_high_res is not None
        hs, ws = (im_A_high_res.shape[-2:] if upsample_res is None else upsample_res)
    low_res_certainty = 0
    if attenuate_cert:
        low = F.interpolate(corresps[16]["certainty"], size=(hs, ws), align_corners=False, mode="bilinear")
        low_res_certainty = 0.5 * low * (low < 0)
    finest = corresps[1]
    if upsample_preds:
        scale_factor = math.sqrt(hs * ws / (560 ** 2))
        corresps = _forward(im_A_high_res, im_B_high_res, sd, dsd, symmetric, True, scale_factor, corresps=finest)
        if stages is not None:
            for s in corresps:
                stages[f"p2_flow{s}"] = corresps[s]["flow"]
                stages[f"p2_cert{s}"] = corresps[s]["certainty"]
    flow = corresps[1]["flow"].permute(0, 2, 3, 1)
    certainty = (corresps[1]["certainty"] - low_res_certainty).sigmoid()
    grid = pixel_grid(b, hs, ws).permute(0, 2, 3, 1)
    if (flow.abs() > 1).any():
        wrong = (flow.abs() > 1).sum(dim=-1) > 0
        certainty[wrong[:, None]] = 0
    flow = torch.clamp(flow, -1, 1)
    if symmetric:
        A_to_B, B_to_A = flow.chunk(2)
        warp = torch.cat((torch.cat((grid, A_to_B), dim=-1), torch.cat((B_to_A, grid), dim=-1)), dim=2)
        certainty = torch.cat(certainty.chunk(2), dim=3)
    else:
        warp = torch.cat((grid, flow), dim=-1)
    return warp, certainty[:, 0]


# --------------------------------------------------------------------------- sampling
def kde(x, std=0.1, half=False, down=None):
    """Gaussian kernel density of matches x [n,4] - romatch/utils/kde.py:4-12.

    The reference computes `(-cdist(x, x[::down])**2 / (2 std^2)).exp().sum(-1)` (in fp16 when half=True).  Restated
    with an explicit squared distance in f32/f64: cdist's sqrt followed by **2 is the identity up to rounding.  With
    half=True only the INPUT rounding of the reference is reproduced (x.half()); the arithmetic stays f32."""
    x = x.detach().to(torch.float32)
    if half:
        x = x.half().float()
    y = x if down is None else x[::down]
    d2 = ((x[:, None, :].double() - y[None, :, :].double()) ** 2).sum(-1)
    return torch.exp(-d2 / (2.0 * std ** 2)).sum(-1).float()


def sample(matches, certainty, num=10000, sample_mode="threshold_balanced", sample_thresh=0.05, generator=None):
    """RegressionMatcher.sample - romatch/models/matcher.py:598-629 (stochastic: two multinomial draws)."""
    if "threshold" in sample_mode:
        certainty = certainty.clone()
        certainty[certainty > sample_thresh] = 1
    matches, certainty = matches.reshape(-1, 4), certainty.reshape(-1)
    expansion_factor = 4 if "balanced" in sample_mode else 1
    good = torch.multinomial(certainty, num_samples=min(expansion_factor * num, len(certainty)), replacement=False,
                             generator=generator)
    good_matches, good_certainty = matches[good], certainty[good]
    if "balanced" not in sample_mode:
        return good_matches, good_certainty
    density = kde(good_matches, std=0.1)
    p = 1 / (density + 1)
    p[density < 10] = 1e-7
    bal = torch.multinomial(p, num_samples=min(num, len(good_certainty)), replacement=False, generator=generator)
    return good_matches[bal], good_certainty[bal]


def match_keypoints(x_A, x_B, warp, certainty, max_dist=0.005, cert_th=0):
    """RegressionMatcher.match_keypoints - romatch/models/matcher.py:732-773; returns (inds_A, inds_B)."""
    x_A_to_B = F.grid_sample(warp[..., -2:].permute(2, 0, 1)[None], x_A[None, None], align_corners=False,
                             mode="bilinear")[0, :, 0].mT
    cert_A_to_B = F.grid_sample(certainty[None, None, ...], x_A[None, None], align_corners=False, mode="bilinear")[0, 0, 0]
    D = torch.cdist(x_A_to_B, x_B)
    return torch.nonzero((D == D.min(dim=-1, keepdim=True).values) * (D == D.min(dim=-2, keepdim=True).values)
                         * (cert_A_to_B[:, None] > cert_th) * (D < max_dist), as_tuple=True)


def visualize_warp(warp, certainty, im_A, im_B, symmetric=True):
    """RegressionMatcher.visualize_warp - romatch/models/matcher.py:964-981 (tensor inputs): [3, H, W2] blend of the
    warped images with the certainty over a white background."""
    H, W2, _ = warp.shape
    W = W2 // 2 if symmetric else W2
    a_rgb = F.grid_sample(im_B[None], warp[:, :W, 2:][None], mode="bilinear", align_corners=False)[0]
    if symmetric:
        b_rgb = F.grid_sample(im_A[None], warp[:, W:, :2][None], mode="bilinear", align_corners=False)[0]
        warp_im = torch.cat((a_rgb, b_rgb), dim=2)
    else:
        warp_im = a_rgb
    return certainty * warp_im + (1 - certainty) * torch.ones((H, W2))


def conf_from_fb_consistency(flow_forward, flow_backward, th=2):
    """RegressionMatcher.conf_from_fb_consistency - romatch/models/matcher.py:672-699 (batched [B,H,W,2] flows)."""
    H, W = flow_forward.shape[-3:-1]
    th_n = 2 * th / max(H, W)
    coords = torch.stack(torch.meshgrid(torch.linspace(-1 + 1 / W, 1 - 1 / W, W), torch.linspace(-1 + 1 / H, 1 - 1 / H, H),
                                        indexing="xy"), dim=-1)
    coords_fb = F.grid_sample(flow_backward.permute(0, 3, 1, 2), flow_forward, align_corners=False, mode="bilinear").permute(0, 2, 3, 1)
    return ((coords - coords_fb).norm(dim=-1) < th_n).float()
