"""Seeded synthetic state-dicts and inputs (no pretrained weights exist offline).

Key / shape contract = the reference's strict `load_state_dict` contract
(romatch/models/model_zoo/roma_models.py:204; DINOv2 dict romatch/models/encoders.py:42-43),
pinned by tests/golden/state_dict_contract.json which was dumped from the reference itself.

Everything is drawn from numpy's PCG64 `Generator` (stream-stable across machines, unlike
vectorised torch CPU normals), so the build container (where goldens are made from the real
reference) and the GPU box (where the HIP path and the oracle run) see bit-identical tensors.

The scales are chosen so that the random network is *numerically well posed* for parity:
  * He/Xavier-like fan-in scaling keeps activations O(1) through ~60 sequential layers;
  * BatchNorm running stats / affine are non-trivial (so BN folding is actually exercised);
  * `to_out` is amplified so the 4096-way class logits are peaked: default-init weights give a
    nearly flat softmax whose arg-max (romatch/utils/utils.py:315) flips on 1e-8 perturbations
    (hard part 1).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

VGG_CONV_IDX = [0, 3, 7, 10, 14, 17, 20, 23, 27, 30, 33, 36]
VGG_CHANNELS = [64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512]

# (in_dim == hidden_dim, displacement_emb_dim, local_corr_radius) per refiner scale
# romatch/models/model_zoo/roma_models.py:103-139
REFINER_CFG = OrderedDict([
    ("16", dict(feat=512, emb=128, radius=7)),
    ("8", dict(feat=512, emb=64, radius=3)),
    ("4", dict(feat=256, emb=32, radius=2)),
    ("2", dict(feat=64, emb=16, radius=0)),
    ("1", dict(feat=9, emb=6, radius=0)),
])
# proj heads: scale -> (cin, cout); roma_models.py:156-160
PROJ_CFG = OrderedDict([("16", (1024, 512)), ("8", (512, 512)), ("4", (256, 256)),
                        ("2", (128, 64)), ("1", (64, 9))])


def refiner_dim(scale: str) -> int:
    c = REFINER_CFG[scale]
    k = (2 * c["radius"] + 1) ** 2 if c["radius"] else 0
    return 2 * c["feat"] + c["emb"] + k


class _Rng:
    def __init__(self, seed: int):
        self.g = np.random.Generator(np.random.PCG64(seed))

    def normal(self, shape, std=1.0):
        a = self.g.standard_normal(size=shape, dtype=np.float32)
        if std != 1.0:
            a *= np.float32(std)
        return torch.from_numpy(a)

    def uniform(self, shape, lo, hi):
        a = self.g.random(size=shape, dtype=np.float32)
        a = a * np.float32(hi - lo) + np.float32(lo)
        return torch.from_numpy(a)


def _bn(sd, rng, prefix, c):
    sd[prefix + ".weight"] = rng.uniform((c,), 0.8, 1.2)
    sd[prefix + ".bias"] = rng.normal((c,), 0.1)
    sd[prefix + ".running_mean"] = rng.normal((c,), 0.1)
    sd[prefix + ".running_var"] = rng.uniform((c,), 0.8, 1.25)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(1000, dtype=torch.int64)


def make_matcher_state_dict(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """603 tensors / 111.36 M parameters."""
    rng = _Rng(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    # --- VGG19-BN features[:40]  (romatch/models/encoders.py:13)
    cin = 3
    for idx, cout in zip(VGG_CONV_IDX, VGG_CHANNELS):
        sd[f"encoder.cnn.layers.{idx}.weight"] = rng.normal((cout, cin, 3, 3), math.sqrt(2.0 / (9 * cin)))
        sd[f"encoder.cnn.layers.{idx}.bias"] = rng.normal((cout,), 0.05)
        _bn(sd, rng, f"encoder.cnn.layers.{idx + 1}", cout)
        cin = cout
    # --- transformer decoder (roma_models.py:75-84): 5 x Block(1024, 8 heads), no qkv bias
    D = 1024
    for i in range(5):
        p = f"decoder.embedding_decoder.blocks.{i}"
        sd[p + ".norm1.weight"] = rng.uniform((D,), 0.8, 1.2)
        sd[p + ".norm1.bias"] = rng.normal((D,), 0.05)
        sd[p + ".attn.qkv.weight"] = rng.normal((3 * D, D), 1.5 / math.sqrt(D))
        sd[p + ".attn.proj.weight"] = rng.normal((D, D), 0.5 / math.sqrt(D))
        sd[p + ".attn.proj.bias"] = rng.normal((D,), 0.02)
        sd[p + ".norm2.weight"] = rng.uniform((D,), 0.8, 1.2)
        sd[p + ".norm2.bias"] = rng.normal((D,), 0.05)
        sd[p + ".mlp.fc1.weight"] = rng.normal((4 * D, D), 1.0 / math.sqrt(D))
        sd[p + ".mlp.fc1.bias"] = rng.normal((4 * D,), 0.02)
        sd[p + ".mlp.fc2.weight"] = rng.normal((D, 4 * D), 0.5 / math.sqrt(4 * D))
        sd[p + ".mlp.fc2.bias"] = rng.normal((D,), 0.02)
    # peaked class logits (see module docstring)
    w_out = rng.normal((64 * 64 + 1, D), 0.25)
    w_out[-1] *= 0.1  # certainty logit row: keep sigmoid(certainty) in its sensitive range
    sd["decoder.embedding_decoder.to_out.weight"] = w_out
    sd["decoder.embedding_decoder.to_out.bias"] = rng.normal((64 * 64 + 1,), 0.1)
    # --- GP (matcher.py:222): Conv2d(2, 512, 1)
    sd["decoder.gps.16.pos_conv.weight"] = rng.normal((512, 2, 1, 1), 0.5)
    sd["decoder.gps.16.pos_conv.bias"] = rng.normal((512,), 0.5)
    # --- proj heads
    for s, (ci, co) in PROJ_CFG.items():
        sd[f"decoder.proj.{s}.0.weight"] = rng.normal((co, ci, 1, 1), 1.0 / math.sqrt(ci))
        sd[f"decoder.proj.{s}.0.bias"] = rng.normal((co,), 0.05)
        _bn(sd, rng, f"decoder.proj.{s}.1", co)
    # --- ConvRefiners (matcher.py:92-122; roma_models.py:85-139)
    for s in REFINER_CFG:
        C = refiner_dim(s)
        p = f"decoder.conv_refiner.{s}"

        def block(bp):
            sd[bp + ".0.weight"] = rng.normal((C, 1, 5, 5), math.sqrt(2.0 / 25.0) * 0.7)
            sd[bp + ".0.bias"] = rng.normal((C,), 0.05)
            _bn(sd, rng, bp + ".1", C)
            sd[bp + ".3.weight"] = rng.normal((C, C, 1, 1), math.sqrt(2.0 / C) * 0.7)
            sd[bp + ".3.bias"] = rng.normal((C,), 0.05)

        block(p + ".block1")
        for hb in range(8):
            block(p + f".hidden_blocks.{hb}")
        sd[p + ".out_conv.weight"] = rng.normal((3, C, 1, 1), 1.0 / math.sqrt(C))
        sd[p + ".out_conv.bias"] = rng.normal((3,), 0.05)
        e = REFINER_CFG[s]["emb"]
        sd[p + ".disp_emb.weight"] = rng.normal((e, 2, 1, 1), 0.7)
        sd[p + ".disp_emb.bias"] = rng.normal((e,), 0.1)
    return sd


def make_dinov2_state_dict(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """343 tensors / 304.37 M parameters: DINOv2 ViT-L/14 (transformer/dinov2.py:333-343)."""
    rng = _Rng(seed + 7919)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    D = 1024
    sd["cls_token"] = rng.normal((1, 1, D), 0.5)
    sd["pos_embed"] = rng.normal((1, 1 + 37 * 37, D), 0.5)
    sd["mask_token"] = torch.zeros(1, D)
    sd["patch_embed.proj.weight"] = rng.normal((D, 3, 14, 14), 1.0 / math.sqrt(3 * 14 * 14))
    sd["patch_embed.proj.bias"] = rng.normal((D,), 0.05)
    for i in range(24):
        p = f"blocks.{i}"
        sd[p + ".norm1.weight"] = rng.uniform((D,), 0.8, 1.2)
        sd[p + ".norm1.bias"] = rng.normal((D,), 0.05)
        sd[p + ".attn.qkv.weight"] = rng.normal((3 * D, D), 1.5 / math.sqrt(D))
        sd[p + ".attn.qkv.bias"] = rng.normal((3 * D,), 0.05)
        sd[p + ".attn.proj.weight"] = rng.normal((D, D), 1.0 / math.sqrt(D))
        sd[p + ".attn.proj.bias"] = rng.normal((D,), 0.02)
        sd[p + ".ls1.gamma"] = rng.uniform((D,), 0.05, 0.3)
        sd[p + ".norm2.weight"] = rng.uniform((D,), 0.8, 1.2)
        sd[p + ".norm2.bias"] = rng.normal((D,), 0.05)
        sd[p + ".mlp.fc1.weight"] = rng.normal((4 * D, D), 1.0 / math.sqrt(D))
        sd[p + ".mlp.fc1.bias"] = rng.normal((4 * D,), 0.02)
        sd[p + ".mlp.fc2.weight"] = rng.normal((D, 4 * D), 1.0 / math.sqrt(4 * D))
        sd[p + ".mlp.fc2.bias"] = rng.normal((D,), 0.02)
        sd[p + ".ls2.gamma"] = rng.uniform((D,), 0.05, 0.3)
    sd["norm.weight"] = rng.uniform((D,), 0.8, 1.2)
    sd["norm.bias"] = rng.normal((D,), 0.05)
    return sd


def make_inputs(batch: int, coarse_res, upsample_res=None, seed: int = 1):
    """Standard-normal images exactly as the reference's timing script feeds them
    (tests/test_roma_upsample_inference_time.py:9-12), but from the portable generator."""
    rng = _Rng(seed)
    ch, cw = (coarse_res, coarse_res) if isinstance(coarse_res, int) else coarse_res
    out = {"im_A": rng.normal((batch, 3, ch, cw)), "im_B": rng.normal((batch, 3, ch, cw))}
    if upsample_res is not None:
        uh, uw = (upsample_res, upsample_res) if isinstance(upsample_res, int) else upsample_res
        out["im_A_high_res"] = rng.normal((batch, 3, uh, uw))
        out["im_B_high_res"] = rng.normal((batch, 3, uh, uw))
    return out


# ------------------------------------------------------------------------------------------------ Tiny RoMa
class XFeatStandIn(torch.nn.Module):
    """A caller-side stand-in for the XFeat backbone TinyRoMa uses (romatch/models/tiny.py:80-99; the real one is
    `torch.hub.load("verlab/accelerated_features", "XFeat").net`, model_zoo/__init__.py:24-27 - not available offline).
    Same interface and the same tensor contract: `norm`, `block1..5`, `skip1`, `block_fusion`;
    x2 = block2(block1(x) + skip1(x)) is [B, 24, H/4, W/4], block_fusion(x3 + x4 + x5) is [B, 64, H/8, W/8].
    Seeded weights (numpy PCG64).  TinyRoMa.__init__ deletes heatmap_head / keypoint_head / fine_matcher, so they exist."""

    def __init__(self, seed: int = 0):
        super().__init__()
        nn = torch.nn
        self.norm = nn.InstanceNorm2d(1)
        self.skip1 = nn.Sequential(nn.AvgPool2d(4, stride=4), nn.Conv2d(1, 24, 1, stride=1, padding=0))
        self.block1 = nn.Sequential(nn.Conv2d(1, 8, 3, stride=2, padding=1), nn.ReLU(), nn.Conv2d(8, 24, 3, stride=2, padding=1), nn.ReLU())
        self.block2 = nn.Sequential(nn.Conv2d(24, 24, 3, padding=1), nn.ReLU())
        self.block3 = nn.Sequential(nn.Conv2d(24, 64, 3, stride=2, padding=1), nn.ReLU())
        self.block4 = nn.Sequential(nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.ReLU())
        self.block5 = nn.Sequential(nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.ReLU())
        self.block_fusion = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 1))
        self.heatmap_head = nn.Identity()
        self.keypoint_head = nn.Identity()
        self.fine_matcher = nn.Identity()
        rng = _Rng(9000 + seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                    m.weight.copy_(rng.normal(tuple(m.weight.shape), std=math.sqrt(2.0 / fan_in)))
                    m.bias.copy_(rng.normal(tuple(m.bias.shape), std=0.1))
        self.train(False)


class XFeatArch(torch.nn.Module):
    """The layer list of the real XFeat backbone (verlab/accelerated_features modules/model.py, XFeatModel - restated from
    the published model definition, the hub checkpoint itself is unobtainable offline): every block is a stack of
    BasicLayer = Conv2d(bias=False) + BatchNorm2d(affine=False) + ReLU (the same BasicLayer as romatch/models/tiny.py:14-28),
    skip1 = AvgPool2d(4, 4) + Conv2d(1, 24, 1), block_fusion ends in a plain 1 x 1 convolution.  Seeded weights and
    non-trivial BatchNorm running statistics (numpy PCG64), eval mode.  Exercises everything the device replay of the
    backbone has to handle: stride-2 and 1 x 1 / padding-0 layers, BatchNorm folding, 1 -> 4 ... 128 -> 128 channels."""

    def __init__(self, seed: int = 0):
        super().__init__()
        nn = torch.nn

        class BasicLayer(nn.Module):
            def __init__(self, cin, cout, kernel_size=3, stride=1, padding=1):
                super().__init__()
                self.layer = nn.Sequential(nn.Conv2d(cin, cout, kernel_size, padding=padding, stride=stride, bias=False),
                                           nn.BatchNorm2d(cout, affine=False), nn.ReLU(inplace=True))

            def forward(self, x):
                return self.layer(x)

        B = BasicLayer
        self.norm = nn.InstanceNorm2d(1)
        self.skip1 = nn.Sequential(nn.AvgPool2d(4, stride=4), nn.Conv2d(1, 24, 1, stride=1, padding=0))
        self.block1 = nn.Sequential(B(1, 4, stride=1), B(4, 8, stride=2), B(8, 8, stride=1), B(8, 24, stride=2))
        self.block2 = nn.Sequential(B(24, 24, stride=1), B(24, 24, stride=1))
        self.block3 = nn.Sequential(B(24, 64, stride=2), B(64, 64, stride=1), B(64, 64, 1, padding=0))
        self.block4 = nn.Sequential(B(64, 64, stride=2), B(64, 64, stride=1), B(64, 64, stride=1))
        self.block5 = nn.Sequential(B(64, 128, stride=2), B(128, 128, stride=1), B(128, 128, stride=1), B(128, 64, 1, padding=0))
        self.block_fusion = nn.Sequential(B(64, 64, stride=1), B(64, 64, stride=1), nn.Conv2d(64, 64, 1, padding=0))
        self.heatmap_head = nn.Identity()
        self.keypoint_head = nn.Identity()
        self.fine_matcher = nn.Identity()
        rng = _Rng(9300 + seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                    m.weight.copy_(rng.normal(tuple(m.weight.shape), std=math.sqrt(2.0 / fan_in)))
                    if m.bias is not None:
                        m.bias.copy_(rng.normal(tuple(m.bias.shape), std=0.1))
                elif isinstance(m, nn.BatchNorm2d):
                    m.running_mean.copy_(rng.normal(tuple(m.running_mean.shape), std=0.2))
                    m.running_var.copy_(rng.uniform(tuple(m.running_var.shape), 0.5, 1.5))
        self.train(False)


def make_tiny_state_dict(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Matcher weights of TinyRoMa (tiny.py:49-62): 4 x BasicLayer (3x3 conv without bias, BatchNorm affine=False) + a 1x1
    conv with bias, for the coarse (130 -> 256 -> 3) and the fine (50 -> 64 -> 3) matcher; same keys as the reference."""
    rng = _Rng(9100 + seed)
    sd = OrderedDict()
    for name, cin, dim in (("coarse_matcher", 64 + 64 + 2, 256), ("fine_matcher", 24 + 24 + 2, 64)):
        c = cin
        for i in range(4):
            sd[f"{name}.{i}.layer.0.weight"] = rng.normal((dim, c, 3, 3), std=math.sqrt(2.0 / (9 * c)))
            sd[f"{name}.{i}.layer.1.running_mean"] = rng.normal((dim,), std=0.2)
            sd[f"{name}.{i}.layer.1.running_var"] = rng.uniform((dim,), 0.5, 1.5)
            sd[f"{name}.{i}.layer.1.num_batches_tracked"] = torch.tensor(100, dtype=torch.long)
            c = dim
        sd[f"{name}.4.weight"] = rng.normal((3, dim, 1, 1), std=0.3 * math.sqrt(1.0 / dim))
        sd[f"{name}.4.bias"] = rng.normal((3,), std=0.05)
    return sd


def make_tiny_inputs(batch: int, h: int, w: int, seed: int = 1):
    """Seeded image pairs in [0, 1] (what ToTensor() yields in TinyRoMa.match_from_path, tiny.py:199-203), smooth enough to
    give structured correlation volumes."""
    rng = _Rng(9200 + seed)
    a = rng.uniform((batch, 3, h, w), 0.0, 1.0)
    b = torch.roll(a, shifts=(h // 9, -(w // 11)), dims=(2, 3)) * 0.8 + 0.2 * rng.uniform((batch, 3, h, w), 0.0, 1.0)
    k = torch.ones(1, 1, 5, 5) / 25.0
    sm = lambda t: torch.nn.functional.conv2d(t.reshape(-1, 1, h, w), k, padding=2).reshape(batch, 3, h, w)  # noqa: E731
    return {"im_A": sm(a).contiguous(), "im_B": sm(b).contiguous()}
