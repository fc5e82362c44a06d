"""CPU restatement of TinyRoMa's inference path (romatch/models/tiny.py) - TEST INFRASTRUCTURE ONLY.

Only tests/, tools/make_goldens.py and bench tooling may import this module; the product (roma_amd/tiny.py) never does.
Pinned against the unmodified reference by tests/golden/tiny_reference.npz (tools/make_goldens.py tinyroma: the
reference's own TinyRoMa with the seeded stand-in backbone roma_amd.synthetic.XFeatStandIn - the real XFeat is an
un-vendored torch.hub dependency, model_zoo/__init__.py:24-27).  Functional torch fp32, every function citing the lines it
follows."""
import math

import torch
import torch.nn.functional as F


def preprocess_tensor(x):
    """tiny.py:71-78: resize so that both sides are multiples of 32."""
    H, W = x.shape[-2:]
    _H, _W = (H // 32) * 32, (W // 32) * 32
    return F.interpolate(x, (_H, _W), mode="bilinear", align_corners=False)


def forward_single(xfeat, x):
    """tiny.py:80-99: grey-scale, instance norm, XFeat pyramid -> (fine feats x2 [B,24,H/4,W/4], coarse feats [B,64,H/8,W/8])."""
    x = x.mean(dim=1, keepdim=True)
    x = xfeat.norm(x)
    x1 = xfeat.block1(x)
    x2 = xfeat.block2(x1 + xfeat.skip1(x))
    x3 = xfeat.block3(x2)
    x4 = xfeat.block4(x3)
    x5 = xfeat.block5(x4)
    x4 = F.interpolate(x4, (x3.shape[-2], x3.shape[-1]), mode="bilinear")
    x5 = F.interpolate(x5, (x3.shape[-2], x3.shape[-1]), mode="bilinear")
    return x2, xfeat.block_fusion(x3 + x4 + x5)


def corr_volume(feat0, feat1):
    """tiny.py:182-196."""
    B, C, H0, W0 = feat0.shape
    _, _, H1, W1 = feat1.shape
    return torch.einsum("bci,bcj->bji", feat0.reshape(B, C, H0 * W0), feat1.reshape(B, C, H1 * W1)).reshape(B, H1, W1, H0, W0) / math.sqrt(C)


def pos_embed(cv, exact_softmax=False):
    """tiny.py:114-142.  exact_softmax=True (:139-141): softmax over every position of image B, expectation of the grid.
    Default = the inference branch (not training, exact_softmax False): low-resolution softmax plus the arg-max
    entry.  Note tiny.py:134 concatenates the arg-max INDEX tensor, so the last logit is the index value itself.

    Batch semantics: tiny.py:137 multiplies P_lowres[:, -1] ([B, H0, W0]) with grid[best_match].permute(0, 3, 1, 2)
    ([B, 2, H0, W0]) - the batch axis of P broadcasts against the CHANNEL axis of the grid, which is only well defined for
    B = 1 (B = 2 silently mixes the two pairs, B >= 3 raises).  The reference's own callers pass single pairs
    (match_from_path / PIL inputs, demo/demo_match_tiny.py); pairs are treated independently here, i.e. every pair gets the
    reference's B = 1 result."""
    if cv.shape[0] > 1:
        return torch.cat([pos_embed(cv[b:b + 1], exact_softmax) for b in range(cv.shape[0])], dim=0)
    B, H1, W1, H0, W0 = cv.shape
    grid = torch.stack(torch.meshgrid(torch.linspace(-1 + 1 / W1, 1 - 1 / W1, W1), torch.linspace(-1 + 1 / H1, 1 - 1 / H1, H1),
                                      indexing="xy"), dim=-1).float().reshape(H1 * W1, 2)
    if exact_softmax:
        P = cv.reshape(B, H1 * W1, H0, W0).softmax(dim=1)
        return torch.einsum("bchw,cd->bdhw", P, grid)
    down = 4
    grid_lr = torch.stack(torch.meshgrid(torch.linspace(-1 + down / W1, 1 - down / W1, W1 // down),
                                         torch.linspace(-1 + down / H1, 1 - down / H1, H1 // down), indexing="xy"),
                          dim=-1).float().reshape(H1 * W1 // down ** 2, 2)
    best_match = cv.reshape(B, H1 * W1, H0, W0).argmax(dim=1)
    P_lowres = torch.cat((cv[:, ::down, ::down].reshape(B, H1 * W1 // down ** 2, H0, W0), best_match[:, None]), dim=1).softmax(dim=1)
    pos = torch.einsum("bchw,cd->bdhw", P_lowres[:, :-1], grid_lr)
    pos = pos + P_lowres[:, -1] * grid[best_match].permute(0, 3, 1, 2)
    return pos


def matcher(x, sd, name):
    """tiny.py:49-62 in eval mode: 4 x (conv3x3 without bias, BatchNorm2d(affine=False) on running statistics, ReLU), 1x1 conv."""
    for i in range(4):
        x = F.conv2d(x, sd[f"{name}.{i}.layer.0.weight"], None, padding=1)
        x = F.batch_norm(x, sd[f"{name}.{i}.layer.1.running_mean"], sd[f"{name}.{i}.layer.1.running_var"], None, None, False, 0.1, 1e-5)
        x = F.relu(x)
    return F.conv2d(x, sd[f"{name}.4.weight"], sd[f"{name}.4.bias"])


def forward_from_features(f0_f, f0_c, f1_f, f1_c, sd, H1, W1, exact_softmax=False):
    """tiny.py:278-303 after forward_single; (H1, W1) = the pre-processed size of image B.  Returns {8: ..., 4: ...}."""
    to_normalized = torch.tensor((2 / W1, 2 / H1, 1.0))[None, :, None, None]
    cv = corr_volume(f0_c, f1_c)
    coarse_warp = pos_embed(cv, exact_softmax)
    coarse_matches = torch.cat((coarse_warp, torch.zeros_like(coarse_warp[:, -1:])), dim=1)
    f1c_w = F.grid_sample(f1_c, coarse_matches.permute(0, 2, 3, 1)[..., :2], mode="bilinear", align_corners=False)
    delta = matcher(torch.cat((f0_c, f1c_w, coarse_warp), dim=1), sd, "coarse_matcher")
    coarse_matches = coarse_matches + delta * to_normalized
    out = {8: {"flow": coarse_matches[:, :2], "certainty": coarse_matches[:, 2:]}}
    up = F.interpolate(coarse_matches, size=f0_f.shape[-2:], mode="bilinear", align_corners=False)
    f1f_w = F.grid_sample(f1_f, up.permute(0, 2, 3, 1)[..., :2], mode="bilinear", align_corners=False)
    fdelta = matcher(torch.cat((f0_f, f1f_w, up[:, :2]), dim=1), sd, "fine_matcher")
    fine = up + fdelta * to_normalized
    out[4] = {"flow": fine[:, :2], "certainty": fine[:, 2:]}
    return out


def forward(im0, im1, xfeat, sd, exact_softmax=False):
    """tiny.py:267-303."""
    im0, im1 = preprocess_tensor(im0), preprocess_tensor(im1)
    f0_f, f0_c = forward_single(xfeat, im0)
    f1_f, f1_c = forward_single(xfeat, im1)
    return forward_from_features(f0_f, f0_c, f1_f, f1_c, sd, im1.shape[-2], im1.shape[-1], exact_softmax)


def finish_match(corresps, H0, W0):
    """tiny.py:222-242: to the resolution of image A, warp = cat(grid, flow), certainty through a sigmoid."""
    B = corresps[4]["flow"].shape[0]
    flow = F.interpolate(corresps[4]["flow"], size=(H0, W0), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).reshape(B, H0, W0, 2)
    grid = torch.stack(torch.meshgrid(torch.linspace(-1 + 1 / W0, 1 - 1 / W0, W0), torch.linspace(-1 + 1 / H0, 1 - 1 / H0, H0),
                                      indexing="xy"), dim=-1).float().expand(B, H0, W0, 2)
    cert = F.interpolate(corresps[4]["certainty"], size=(H0, W0), mode="bilinear", align_corners=False)
    return torch.cat((grid, flow), dim=-1), cert[:, 0].sigmoid()


def match(im0, im1, xfeat, sd, exact_softmax=False):
    """TinyRoMa.match for batched tensors (tiny.py:205-242)."""
    with torch.no_grad():
        return finish_match(forward(im0, im1, xfeat, sd, exact_softmax), im0.shape[-2], im0.shape[-1])
