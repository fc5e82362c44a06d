"""Tiny RoMa on MI355X: mirror of romatch.models.tiny.TinyRoMa (tiny.py:30-303) for inference.

The XFeat backbone is an un-vendored torch.hub dependency of the reference (model_zoo/__init__.py:24-27:
`torch.hub.load("verlab/accelerated_features", "XFeat").net`), so - exactly as there - the caller passes it in as `xfeat=`
and it runs as the caller's torch module (tiny.py:80-99).  Everything after the backbone is hand-written HIP behind the
C ABI (include/roma_hip.h, csrc/tiny.hip): the all-pairs correlation volume on the MFMA GEMM, the soft arg-max position
embedding, the two convolutional matchers (implicit-GEMM 3x3 convolutions with the BatchNorm folded in), the grid-sample
warps and the final assembly.  fp32 throughout (the reference runs TinyRoMa in fp32).  No CPU fallback: tensors must live on
a HIP device.

Batches: pairs are independent here, each gets the reference's single-pair (B = 1) result.  The reference's own batched
pos_embed broadcasts the batch axis against the channel axis (tiny.py:137) - B = 2 silently mixes the pairs, B >= 3 raises -
and its callers pass single pairs (match_from_path, demo/demo_match_tiny.py)."""
from __future__ import annotations

import ctypes as C
import math
from pathlib import Path

import numpy as np
import torch

from . import _lib
from .kde import kde
from .sampling import multinomial

F32 = 0


def _P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _pack_conv3x3(w: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, cin_p: int, eps: float = 1e-5):
    """conv3x3 (no bias) + BatchNorm2d(affine=False) on running statistics (tiny.py:19-24) -> weights [Cout][ky][kx][cin_p]
    (zero-padded input channels) and a bias, for the NHWC implicit-GEMM convolution."""
    inv = torch.rsqrt(var.double() + eps)
    wf = (w.double() * inv[:, None, None, None]).permute(0, 2, 3, 1)  # [Cout, 3, 3, Cin]
    cout, _, _, cin = wf.shape
    out = torch.zeros((cout, 3, 3, cin_p), dtype=torch.float64)
    out[..., :cin] = wf
    return out.reshape(cout, 9 * cin_p).float().contiguous(), (-mean.double() * inv).float().contiguous()


# ---------------------------------------------------------------------------------------------------- backbone compiler
def _leaves(m):
    """Leaf modules of a Sequential-like tree in execution order (containers whose forward is "run the children in
    registration order": nn.Sequential, XFeat's BasicLayer whose only child is a Sequential)."""
    kids = list(m.children())
    if not kids:
        yield m
        return
    for k in kids:
        yield from _leaves(k)


def _compile_stack(m, dev):
    """One XFeat block -> a list of device ops.  Every Conv2d absorbs the BatchNorm2d (running statistics, affine or not)
    and the ReLU that follow it; weights are re-laid-out to [K*K*Cin][Cout] for roma_op_conv2d_nhwc.  Anything that is
    not Conv2d / BatchNorm2d / ReLU / AvgPool2d / Identity is refused (no silent fallback to the torch module)."""
    nn = torch.nn
    ops = []
    for leaf in _leaves(m):
        if isinstance(leaf, nn.Conv2d):
            k, st, pd = leaf.kernel_size, leaf.stride, leaf.padding
            if (k[0] != k[1] or k[0] not in (1, 3) or st[0] != st[1] or st[0] not in (1, 2) or pd[0] != pd[1] or pd[0] not in (0, 1)
                    or leaf.dilation != (1, 1) or leaf.groups != 1 or leaf.padding_mode != "zeros" or leaf.out_channels % 4):
                raise NotImplementedError(f"TinyRoMa backbone: unsupported convolution {leaf}")
            w = leaf.weight.detach().double().cpu()
            b = leaf.bias.detach().double().cpu() if leaf.bias is not None else torch.zeros(leaf.out_channels, dtype=torch.float64)
            ops.append(dict(op="conv", w=w, b=b, k=k[0], s=st[0], p=pd[0], relu=0, cin=leaf.in_channels, cout=leaf.out_channels))
        elif isinstance(leaf, nn.BatchNorm2d):
            if not ops or ops[-1]["op"] != "conv" or ops[-1]["relu"]:
                raise NotImplementedError("TinyRoMa backbone: BatchNorm2d must directly follow a convolution")
            inv = torch.rsqrt(leaf.running_var.detach().double().cpu() + leaf.eps)
            g = leaf.weight.detach().double().cpu() * inv if leaf.affine else inv
            sh = leaf.bias.detach().double().cpu() if leaf.affine else torch.zeros_like(inv)
            c = ops[-1]
            c["w"] = c["w"] * g[:, None, None, None]
            c["b"] = (c["b"] - leaf.running_mean.detach().double().cpu()) * g + sh
        elif isinstance(leaf, nn.ReLU):
            if not ops or ops[-1]["op"] != "conv":
                raise NotImplementedError("TinyRoMa backbone: ReLU must follow a convolution")
            ops[-1]["relu"] = 1
        elif isinstance(leaf, nn.AvgPool2d):
            k = leaf.kernel_size if isinstance(leaf.kernel_size, int) else leaf.kernel_size[0]
            st = leaf.stride if isinstance(leaf.stride, int) else leaf.stride[0]
            if k != st or leaf.padding not in (0, (0, 0)):
                raise NotImplementedError(f"TinyRoMa backbone: unsupported pooling {leaf}")
            ops.append(dict(op="avgpool", k=k))
        elif isinstance(leaf, nn.Identity):
            continue
        else:
            raise NotImplementedError(f"TinyRoMa backbone: no device operator for {type(leaf).__name__}")
    for c in ops:
        if c["op"] == "conv":  # [Cout, Cin, K, K] -> [(ky K + kx) Cin + ci][Cout]
            c["w"] = c["w"].permute(2, 3, 1, 0).reshape(-1, c["cout"]).float().contiguous().to(dev)
            c["b"] = c["b"].float().contiguous().to(dev)
    return ops


class TinyRoMa:
    """Same constructor arguments and public methods as the reference class (tiny.py:36-66, 101-112, 144-180, 198-265)."""

    def __init__(self, xfeat=None, freeze_xfeat=True, sample_mode="threshold_balanced", symmetric=False, exact_softmax=False,
                 weights=None, device="cuda:0"):
        if xfeat is None:
            raise ValueError("TinyRoMa: pass the XFeat backbone as xfeat= (the reference loads it from torch.hub; there is no "
                             "network here)")
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise _lib.RomaHipError("roma_amd.TinyRoMa needs a HIP device; there is no CPU fallback")
        for name in ("heatmap_head", "keypoint_head", "fine_matcher"):  # tiny.py:43
            if hasattr(xfeat, name):
                delattr(xfeat, name)
        self.xfeat = [xfeat.train(False)]
        # the backbone runs on the device through roma_op_* (forward_single): the module is walked once, here
        nrm = getattr(xfeat, "norm", None)
        if not isinstance(nrm, torch.nn.InstanceNorm2d) or nrm.affine or nrm.track_running_stats or nrm.num_features != 1:
            raise NotImplementedError("TinyRoMa backbone: xfeat.norm must be InstanceNorm2d(1) (no affine, no running statistics)")
        self._norm_eps = float(nrm.eps)
        self._prog = {name: _compile_stack(getattr(xfeat, name), self._device)
                      for name in ("skip1", "block1", "block2", "block3", "block4", "block5", "block_fusion")}
        self.freeze_xfeat = freeze_xfeat
        self.sample_mode = sample_mode
        self.sample_thresh = 0.05
        self.symmetric = symmetric
        self.exact_softmax = exact_softmax
        self._lib = _lib.load()
        self._w = None
        if weights is not None:
            self.load_state_dict(weights)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=True):
        """Keys of the reference state dict (tiny.py:49-62); `xfeat.*` entries (present when the reference model was built
        with freeze_xfeat=False) are loaded into the caller's backbone."""
        need = []
        for name in ("coarse_matcher", "fine_matcher"):
            for i in range(4):
                need += [f"{name}.{i}.layer.0.weight", f"{name}.{i}.layer.1.running_mean", f"{name}.{i}.layer.1.running_var"]
            need += [f"{name}.4.weight", f"{name}.4.bias"]
        missing = [k for k in need if k not in sd]
        extra = [k for k in sd if k not in need and not k.endswith("num_batches_tracked") and not k.startswith("xfeat.")]
        if strict and (missing or extra):
            raise RuntimeError(f"TinyRoMa.load_state_dict: missing {missing}, unexpected {extra}")
        xs = {k[len("xfeat.0."):]: v for k, v in sd.items() if k.startswith("xfeat.0.")}
        if xs:
            self.xfeat[0].load_state_dict(xs, strict=False)
            self._prog = {name: _compile_stack(getattr(self.xfeat[0], name), self._device) for name in self._prog}
        dev = self._device
        w = {}
        for name, cin, cin_p in (("coarse_matcher", 130, 160), ("fine_matcher", 50, 64)):
            layers = []
            c_p = cin_p
            for i in range(4):
                wt, b = _pack_conv3x3(sd[f"{name}.{i}.layer.0.weight"].cpu(), sd[f"{name}.{i}.layer.1.running_mean"].cpu(),
                                      sd[f"{name}.{i}.layer.1.running_var"].cpu(), c_p)
                layers.append((wt.to(dev), b.to(dev), c_p, int(wt.shape[0])))
                c_p = int(wt.shape[0])
            w[name] = dict(layers=layers, cin_p=cin_p,
                           out_w=sd[f"{name}.4.weight"].cpu().float().reshape(3, -1).contiguous().to(dev),
                           out_b=sd[f"{name}.4.bias"].cpu().float().contiguous().to(dev))
        self._w = w
        return self

    @property
    def device(self):
        return self._device

    # ------------------------------------------------------------------ backbone on the device (tiny.py:71-99)
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    def _run_stack(self, name, x, res_last=None):
        """x [B,H,W,C] channels-last f32 through the compiled layers of xfeat.<name>; `res_last` is added to the output of
        the last layer (x1 + skip1(x), tiny.py:89)."""
        lib, dev, stream = self._lib, self._device, self._stream()
        ops = self._prog[name]
        for n, o in enumerate(ops):
            B, H, W, Cc = x.shape
            if o["op"] == "avgpool":
                out = torch.empty((B, H // o["k"], W // o["k"], Cc), device=dev, dtype=torch.float32)
                _lib.check(lib.roma_op_avgpool_nhwc(_P(x), _P(out), B, H, W, Cc, o["k"], stream))
            else:
                if Cc != o["cin"]:
                    raise RuntimeError(f"TinyRoMa backbone: {name} expects {o['cin']} input channels, got {Cc}")
                Ho, Wo = (H + 2 * o["p"] - o["k"]) // o["s"] + 1, (W + 2 * o["p"] - o["k"]) // o["s"] + 1
                out = torch.empty((B, Ho, Wo, o["cout"]), device=dev, dtype=torch.float32)
                res = res_last if n == len(ops) - 1 else None
                if res is not None and tuple(res.shape) != tuple(out.shape):
                    raise RuntimeError("TinyRoMa backbone: skip connection and block output differ in shape")
                _lib.check(lib.roma_op_conv2d_nhwc(_P(x), _P(o["w"]), _P(o["b"]), _P(res), _P(out), B, H, W, Cc, o["cout"], o["k"],
                                                   o["s"], o["p"], o["relu"], stream))
            x = out
        return x

    def _images_nhwc(self, x):
        """NCHW image batch -> channels-last f32 on the device, resized to multiples of 32 (preprocess_tensor, tiny.py:72-79)."""
        x = x.detach().to(self._device, torch.float32).contiguous()
        B, Cc, H, W = x.shape
        lib, stream = self._lib, self._stream()
        with torch.cuda.device(self._device):
            t = torch.empty((B, H, W, Cc), device=self._device, dtype=torch.float32)
            _lib.check(lib.roma_op_nchw_to_nhwc(_P(x), _P(t), B, Cc, H, W, stream))
            _H, _W = (H // 32) * 32, (W // 32) * 32
            if (_H, _W) != (H, W):
                r = torch.empty((B, _H, _W, Cc), device=self._device, dtype=torch.float32)
                _lib.check(lib.roma_op_resize_bilinear(_P(t), _P(r), B, H, W, _H, _W, Cc, stream))
                t = r
        return t

    def preprocess_tensor(self, x):
        """tiny.py:72-79 (bilinear resize to multiples of 32 on the device); returns (NCHW tensor, rh, rw) like the reference."""
        H, W = x.shape[-2:]
        _H, _W = (H // 32) * 32, (W // 32) * 32
        return self._images_nhwc(x).permute(0, 3, 1, 2), H / _H, W / _W

    def _forward_single_nhwc(self, t):
        """t [B,H,W,C] channels-last pre-processed images -> (x2 [B,H/4,W/4,24], feats [B,H/8,W/8,64]) channels-last."""
        lib, dev = self._lib, self._device
        with torch.cuda.device(dev):
            stream = self._stream()
            B, H, W, Cc = t.shape
            g = torch.empty((B, H, W, 1), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_gray_instnorm(_P(t), _P(g), B, H, W, Cc, self._norm_eps, stream))  # x.mean(1) -> norm
            x1 = self._run_stack("block1", g)
            x2 = self._run_stack("block2", self._run_stack("skip1", g, res_last=x1))  # block2(x1 + skip1(x))
            x3 = self._run_stack("block3", x2)
            x4 = self._run_stack("block4", x3)
            x5 = self._run_stack("block5", x4)
            _, h3, w3, c3 = x3.shape
            ups = []
            for xx in (x4, x5):  # F.interpolate(..., mode="bilinear") to x3's size (tiny.py:94-95)
                u = torch.empty((B, h3, w3, xx.shape[-1]), device=dev, dtype=torch.float32)
                _lib.check(lib.roma_op_resize_bilinear(_P(xx), _P(u), B, xx.shape[1], xx.shape[2], h3, w3, xx.shape[-1], stream))
                ups.append(u)
            sm = torch.empty_like(x3)
            _lib.check(lib.roma_op_add3(_P(x3), _P(ups[0]), _P(ups[1]), _P(sm), sm.numel(), stream))
            feats = self._run_stack("block_fusion", sm)
        return x2, feats

    @torch.no_grad()
    def forward_single(self, x):
        """tiny.py:81-99 on the device; x [B,C,H,W] (already pre-processed) -> (x2, feats) as NCHW views."""
        x = x.detach().to(self._device, torch.float32).contiguous()
        B, Cc, H, W = x.shape
        with torch.cuda.device(self._device):
            t = torch.empty((B, H, W, Cc), device=self._device, dtype=torch.float32)
            _lib.check(self._lib.roma_op_nchw_to_nhwc(_P(x), _P(t), B, Cc, H, W, self._stream()))
        x2, feats = self._forward_single_nhwc(t)
        return x2.permute(0, 3, 1, 2), feats.permute(0, 3, 1, 2)

    # ------------------------------------------------------------------ device side
    def _nhwc(self, x, stream):
        x = x.detach().to(self._device, torch.float32).contiguous()
        B, Cc, H, W = x.shape
        out = torch.empty((B, H, W, Cc), device=self._device, dtype=torch.float32)
        _lib.check(self._lib.roma_op_nchw_to_nhwc(_P(x), _P(out), B, Cc, H, W, stream))
        return out

    def _matcher(self, name, d, B, H, W, stream):
        """4 x (conv3x3 + folded BN + ReLU) and the 1x1 output convolution (tiny.py:49-62) on a channels-last input."""
        cfg = self._w[name]
        cur = d
        for (wt, b, cin_p, cout) in cfg["layers"]:
            nxt = torch.empty((B, H, W, cout), device=self._device, dtype=torch.float32)
            _lib.check(self._lib.roma_op_conv3x3(_P(cur), _P(wt), _P(b), _P(nxt), B, H, W, cin_p, cout, 1, F32, stream))
            cur = nxt
        M, K = B * H * W, cur.shape[-1]
        delta = torch.empty((M, 4), device=self._device, dtype=torch.float32)  # 16-byte rows for the GEMM epilogue
        _lib.check(self._lib.roma_op_gemm(_P(cur), K, _P(cfg["out_w"]), K, _P(delta), 4, M, 3, K, 1, 0, 0, 0, _P(cfg["out_b"]),
                                          None, None, 0, 0, 1.0, F32, F32, stream))
        return delta

    @torch.no_grad()
    def forward_from_features(self, f0_f, f0_c, f1_f, f1_c, H1, W1):
        """tiny.py:278-303 after forward_single, on NCHW torch feature maps; (H1, W1) = pre-processed size of image B.
        Returns the reference's `corresps` dict (NCHW tensors)."""
        if self._w is None:
            raise RuntimeError("TinyRoMa: load_state_dict() first")
        with torch.cuda.device(self._device):
            stream = C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)
            a_c, b_c, a_f, b_f = (self._nhwc(t, stream) for t in (f0_c, f1_c, f0_f, f1_f))
        return self._forward_from_nhwc(a_f, a_c, b_f, b_c, H1, W1)

    def _forward_from_nhwc(self, a_f, a_c, b_f, b_c, H1, W1):
        """the same on channels-last device tensors (what the on-device backbone produces)"""
        lib, dev = self._lib, self._device
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            B, Hc, Wc, Cc = a_c.shape
            _, Hc1, Wc1, _ = b_c.shape
            _, Hf, Wf, Cf = a_f.shape
            _, Hf1, Wf1, _ = b_f.shape
            n0, n1 = Hc * Wc, Hc1 * Wc1
            # corr_volume (tiny.py:182-196): cv[b, j, i] = <f1[b, j], f0[b, i]> / sqrt(C) on the batched f32 MFMA GEMM
            cv = torch.empty((B, n1, n0), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_gemm(_P(b_c), Cc, _P(a_c), Cc, _P(cv), n0, n1, n0, Cc, B, n1 * Cc, n0 * Cc, n1 * n0, None, None, None,
                                        0, 0, 1.0 / math.sqrt(Cc), F32, F32, stream))
            cw = torch.empty((B, Hc, Wc, 2), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_tiny_pos_embed(_P(cv), _P(cw), B, Hc1, Wc1, Hc, Wc, int(bool(self.exact_softmax)), stream))
            # coarse matcher
            cp = self._w["coarse_matcher"]["cin_p"]
            d = torch.empty((B, Hc, Wc, cp), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_tiny_matcher_input(_P(a_c), _P(b_c), _P(cw), 2, _P(d), B, Hc, Wc, Hc1, Wc1, Cc, cp, stream))
            delta = self._matcher("coarse_matcher", d, B, Hc, Wc, stream)
            cm = torch.empty((B, Hc, Wc, 3), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_tiny_update(_P(cw), 2, _P(delta), 4, 2.0 / W1, 2.0 / H1, _P(cm), B * n0, stream))
            # fine matcher on the up-sampled coarse matches (tiny.py:294-300)
            up = torch.empty((B, Hf, Wf, 3), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_resize_bilinear(_P(cm), _P(up), B, Hc, Wc, Hf, Wf, 3, stream))
            fp = self._w["fine_matcher"]["cin_p"]
            df = torch.empty((B, Hf, Wf, fp), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_tiny_matcher_input(_P(a_f), _P(b_f), _P(up), 3, _P(df), B, Hf, Wf, Hf1, Wf1, Cf, fp, stream))
            fdelta = self._matcher("fine_matcher", df, B, Hf, Wf, stream)
            fm = torch.empty((B, Hf, Wf, 3), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_tiny_update(_P(up), 3, _P(fdelta), 4, 2.0 / W1, 2.0 / H1, _P(fm), B * Hf * Wf, stream))
        self._fine_nhwc = fm
        cmc, fmc = cm.permute(0, 3, 1, 2), fm.permute(0, 3, 1, 2)
        return {8: {"flow": cmc[:, :2], "certainty": cmc[:, 2:]}, 4: {"flow": fmc[:, :2], "certainty": fmc[:, 2:]}}

    @torch.no_grad()
    def forward(self, batch):
        """tiny.py:267-303."""
        t0, t1 = self._images_nhwc(batch["im_A"]), self._images_nhwc(batch["im_B"])  # preprocess_tensor (tiny.py:269-270)
        f0_f, f0_c = self._forward_single_nhwc(t0)
        f1_f, f1_c = self._forward_single_nhwc(t1)
        return self._forward_from_nhwc(f0_f, f0_c, f1_f, f1_c, t1.shape[1], t1.shape[2])

    def _finish(self, B, H0, W0):
        """tiny.py:222-242 from the channels-last fine matches of the last forward."""
        lib, dev = self._lib, self._device
        fm = self._fine_nhwc
        _, Hf, Wf, _ = fm.shape
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            full = torch.empty((B, H0, W0, 3), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_resize_bilinear(_P(fm), _P(full), B, Hf, Wf, H0, W0, 3, stream))
            warp = torch.empty((B, H0, W0, 4), device=dev, dtype=torch.float32)
            cert = torch.empty((B, H0, W0), device=dev, dtype=torch.float32)
            _lib.check(lib.roma_op_tiny_final(_P(full), _P(warp), _P(cert), B, H0, W0, stream))
        return warp, cert

    @torch.no_grad()
    def match_from_path(self, im0_path, im1_path):
        from PIL import Image
        to_t = lambda p: torch.from_numpy(np.array(Image.open(p).convert("RGB"))).permute(2, 0, 1).float().div(255)[None]  # noqa: E731
        return self.match(to_t(im0_path).to(self._device), to_t(im1_path).to(self._device), batched=False)

    @torch.no_grad()
    def match(self, im0, im1, *args, batched=True):
        """tiny.py:205-242: (warp [B,H0,W0,4], certainty [B,H0,W0]) at the resolution of im0."""
        from PIL import Image
        if isinstance(im0, (str, Path)):
            return self.match_from_path(im0, im1)
        if isinstance(im0, Image.Image):
            batched = False
            to_t = lambda im: torch.from_numpy(np.array(im.convert("RGB"))).permute(2, 0, 1).float().div(255)[None]  # noqa: E731
            im0, im1 = to_t(im0).to(self._device), to_t(im1).to(self._device)
        B, _, H0, W0 = im0.shape
        self.forward({"im_A": im0, "im_B": im1})
        warp, cert = self._finish(B, H0, W0)
        return (warp, cert) if batched else (warp[0], cert[0])

    # ------------------------------------------------------------------ post-processing shared with RegressionMatcher
    def sample(self, matches, certainty, num=5_000):
        """tiny.py:244-274 (density from the HIP all-pairs KDE)."""
        if "threshold" in self.sample_mode:
            certainty = certainty.clone()
            certainty[certainty > self.sample_thresh] = 1
        matches, certainty = matches.reshape(-1, 4), certainty.reshape(-1)
        expansion_factor = 4 if "balanced" in self.sample_mode else 1
        good = multinomial(certainty, min(expansion_factor * num, len(certainty)))
        good_matches, good_certainty = matches[good], certainty[good]
        if "balanced" not in self.sample_mode:
            return good_matches, good_certainty
        density = kde(good_matches, std=0.1, half=True, down=1)
        p = 1 / (density + 1)
        p[density < 10] = 1e-7
        bal = multinomial(p, min(num, len(good_certainty)))
        return good_matches[bal], good_certainty[bal]

    def to_pixel_coordinates(self, coords, H_A, W_A, H_B=None, W_B=None):
        """tiny.py:101-112."""
        if coords.shape[-1] == 2:
            return self._to_pixel_coordinates(coords, H_A, W_A)
        if isinstance(coords, (list, tuple)):
            kpts_A, kpts_B = coords[0], coords[1]
        else:
            kpts_A, kpts_B = coords[..., :2], coords[..., 2:]
        return self._to_pixel_coordinates(kpts_A, H_A, W_A), self._to_pixel_coordinates(kpts_B, H_B, W_B)

    def _to_pixel_coordinates(self, coords, H, W):
        return torch.stack((W / 2 * (coords[..., 0] + 1), H / 2 * (coords[..., 1] + 1)), axis=-1)

    def visualize_warp(self, warp, certainty, im_A=None, im_B=None, im_A_path=None, im_B_path=None, symmetric=True,
                       save_path=None, unnormalize=False):
        """tiny.py:144-180 (same kernel as RegressionMatcher.visualize_warp)."""
        from .matcher import RegressionMatcher
        return RegressionMatcher.visualize_warp(self, warp, certainty, im_A=im_A, im_B=im_B, im_A_path=im_A_path,
                                                im_B_path=im_B_path, symmetric=symmetric, save_path=save_path,
                                                unnormalize=unnormalize)


def tiny_roma_v1_outdoor(device, weights=None, xfeat=None):
    """model_zoo/__init__.py:18-28.  Offline there is neither the weight URL nor torch.hub: both arguments are required."""
    if weights is None or xfeat is None:
        raise ValueError("tiny_roma_v1_outdoor: pass weights= (the tiny_roma_v1_outdoor.pth state dict) and xfeat= "
                         "(torch.hub 'verlab/accelerated_features' XFeat().net); no network access here")
    return TinyRoMa(xfeat=xfeat, freeze_xfeat=False, exact_softmax=False, weights=weights, device=device)
