"""Host-side mirror of the reference's matcher API over libroma_hip (ctypes).

Same names, argument meaning and error behaviour as
  romatch/models/matcher.py:550-934        RegressionMatcher (match / attributes / helpers)
  romatch/models/model_zoo/roma_models.py:32-205   roma_model
  romatch/models/model_zoo/__init__.py:31-93       roma_outdoor / roma_indoor
All arithmetic of match() runs in hand-written HIP kernels; torch is only the tensor
container (device memory, current stream).  There is no CPU path: a missing library or a
non-GPU device raises.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
import math
import os
from typing import Optional, Union
from warnings import warn

import numpy as np
import torch

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _to_hw(res):
    if res is None:
        return None
    if isinstance(res, int):
        return (res, res)
    return (int(res[0]), int(res[1]))


def _check_input(im_input):
    """romatch/models/matcher.py:530-547 (same exceptions)."""
    from PIL import Image
    if isinstance(im_input, (str, os.PathLike)):
        im = Image.open(im_input)
        if im.mode == "I;16":  # utils.py:655-657
            raise NotImplementedError("Can't handle 16 bit images")
        return im.convert("RGB")
    if isinstance(im_input, Image.Image):
        if im_input.mode != "RGB":  # utils.py:659-661
            raise NotImplementedError("Can't handle non-RGB images")
        return im_input
    assert isinstance(im_input, torch.Tensor), "im_input must be a string, path, or PIL image"
    B, Cc, H, W = im_input.shape
    assert Cc == 3, "im_input must be a RGB image"
    assert H % 14 == 0, "im_input must be a multiple of 14"
    assert W % 14 == 0, "im_input must be a multiple of 14"
    return im_input


def _pil_to_normalised(im, hw):
    """get_tuple_transform_ops(resize=hw, normalize=True) (utils/utils.py:164-173): PIL bicubic resize,
    /255, ImageNet mean/std."""
    from PIL import Image
    h, w = hw
    im = im.resize((w, h), resample=Image.BICUBIC)
    a = np.array(im, dtype=np.float32).transpose((2, 0, 1)) / np.float32(255.0)
    mean = np.array(IMAGENET_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(IMAGENET_STD, dtype=np.float32)[:, None, None]
    return torch.from_numpy((a - mean) / std)


class RegressionMatcher:
    """Drop-in for romatch.models.matcher.RegressionMatcher (inference surface)."""

    HANDLE_CACHE = 2  # default number of library handles (resolution configurations) kept alive per matcher

    def __init__(self, weights, dinov2_weights, h=560, w=560, sample_mode="threshold_balanced", upsample_preds=False,
                 symmetric=False, sample_thresh=0.05, name=None, attenuate_cert=None, upsample_res=None,
                 device=None, amp_dtype=torch.float16, max_batch=8, decoder_dtype=None, handle_cache=None):
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise _lib.RomaHipError(f"roma_amd runs only on a HIP device (got device={device!r}); there is no CPU fallback")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.attenuate_cert = attenuate_cert
        self.name = name
        self.w_resized = w
        self.h_resized = h
        self.sample_mode = sample_mode
        self.upsample_preds = upsample_preds
        self.upsample_res = _to_hw(upsample_res) or (14 * 16 * 6, 14 * 16 * 6)
        self.symmetric = symmetric
        self.sample_thresh = sample_thresh
        if amp_dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"amp_dtype must be torch.float32, torch.bfloat16 or torch.float16 (got {amp_dtype})")
        # torch.float16 (the reference's GPU default, model_zoo/__init__.py:37) runs on libroma_hip_f16.so: IEEE binary16
        # storage, f32 accumulate; torch.bfloat16 (the reference's timing script) and torch.float32 on libroma_hip.so.
        # gfx950's MFMA has one 16-bit rate for both formats.
        self.amp_dtype = amp_dtype
        # decoder_dtype: the 16-bit format of everything that is NOT DINOv2.  In the reference `amp_dtype` reaches only the
        # DINOv2 backbone (model_zoo/roma_models.py:183-188); the VGG pyramid, the decoder and the refiners autocast to
        # float16 whatever it is (encoders.py:7, matcher.py:46,341) - so the reference's timing script (amp_dtype=bfloat16)
        # runs bf16 DINOv2 + fp16 elsewhere.  decoder_dtype=torch.float16 with amp_dtype=torch.bfloat16 selects exactly
        # that (ROMA_MIXED: libroma_hip_f16.so runs DINOv2 through libroma_hip.so); None keeps ONE format everywhere.
        if decoder_dtype is not None and decoder_dtype != amp_dtype:
            if not (amp_dtype == torch.bfloat16 and decoder_dtype == torch.float16):
                raise ValueError("decoder_dtype: only amp_dtype=torch.bfloat16 with decoder_dtype=torch.float16 is a mixed mode")
        self.decoder_dtype = decoder_dtype
        self.mixed = decoder_dtype is not None and decoder_dtype != amp_dtype
        self._lib = _lib.load("f16" if self.mixed else _lib.fmt_of(amp_dtype))
        # resolution configurations kept alive: each holds its packed weights (~0.9 GB in 16-bit modes) and a workspace
        # planned for max_batch (~12 GB at batch 8, 560 -> 864); 1 releases the old handle as soon as the new one is built
        self.handle_cache = int(handle_cache) if handle_cache is not None else self.HANDLE_CACHE
        self.max_batch = int(max_batch)
        self.training = False
        self.debug = False
        # bf16 mode only: DINOv2's residual stream in bf16 like the reference's bf16 backbone (encoders.py casts the
        # backbone weights and input to amp_dtype); False keeps it in f32 (slower, slightly closer to the fp32 result)
        self.vit_bf16_residual = True
        # batches of >= 2 pairs run as two half-batches on two HIP streams (+5 % at batch 8; bit-identical to the
        # single-stream schedule: tests/test_gpu_match.py::test_stream_split_*).  False = one stream, half the workspace
        self.dual_stream = True
        self.trace = False  # tests / tools only: per-stage output checksums (debug_trace)
        # the last ConvRefiner block's 1x1 and out_conv evaluated as one composed C -> 3 map (linear o linear; include/roma_hip.h
        # "compose_out_conv"); False = the reference's two steps (results differ by rounding only)
        self.compose_out_conv = os.environ.get("ROMA_COMPOSE_OUT", "1") != "0"
        self._weights = weights
        self._dinov2_weights = dinov2_weights
        self._handle = None
        self._built = None
        # a call whose tensors have another resolution than the configured one needs its own handle (weights re-packed,
        # workspace re-planned: ~1.3 GB of uploads).  The last HANDLE_CACHE configurations are kept, so alternating
        # between two resolutions (e.g. landscape / portrait inputs) rebuilds nothing.
        self._cache = OrderedDict()
        self._ensure_handle()

    # ------------------------------------------------------------------ handle management
    def _config_key(self, hw=None):
        """(coarse h, w, upsample (h, w) or (0, 0), precision, max_batch, device).  `hw` = the resolution of THIS call's
        tensors when it differs from the configured one (the attributes are never mutated, as in the reference)."""
        h, w = hw if hw is not None else (self.h_resized, self.w_resized)
        up = tuple(int(v) for v in self.upsample_res) if self.upsample_preds else (0, 0)
        prec = _lib.ROMA_F32 if self.amp_dtype == torch.float32 else (_lib.ROMA_MIXED if self.mixed else _lib.H16_CODE[self._lib.h16])
        return (int(h), int(w), up, prec, self.max_batch, self.device.index)

    def _ensure_handle(self, hw=None):
        key = self._config_key(hw)
        if self._handle is not None and self._built == key:
            return
        if (self._handle is not None and key[2] == (0, 0) and self._built[:2] + self._built[3:] == key[:2] + key[3:]):
            return  # upsample_preds switched off: the handle planned for the upsample pass also runs coarse-only
        if key in self._cache:  # a configuration used before: switch, no rebuild
            self._cache.move_to_end(key)
            self._handle, self._built = self._cache[key], key
            return
        try:
            h = self._build_handle(key)
        except _lib.RomaHipError:
            raise
        except (RuntimeError, AssertionError) as e:
            # Peak residency while switching is handle_cache + 1 handles (each ~0.9 GB of packed weights + a workspace of
            # ~12 GB at batch 8, 560 -> 864): the new handle is built before the old ones are evicted, so that a failed
            # build leaves the matcher on its old, working handle.  If the build failed on the DEVICE (out of memory in
            # roma_finalize) and older handles are still resident, release them and try once more.
            if not self._cache or "hip" not in str(e).lower():
                raise
            self._release()
            h = self._build_handle(key)
        while len(self._cache) >= max(self.handle_cache, 1):
            _, old = self._cache.popitem(last=False)
            self._lib.roma_destroy(old)
        self._handle = h
        self._built = key
        self._cache[key] = h

    def _build_handle(self, key):
        """roma_create + roma_set_tensor x 946 + roma_finalize for one configuration; raises and destroys on failure."""
        lib = self._lib
        uh, uw = key[2]
        cfg = _lib.RomaConfig(key[0], key[1], uh, uw, int(bool(self.symmetric)), int(bool(self.upsample_preds)),
                              int(bool(self.attenuate_cert)), key[3], self.max_batch, self.device.index)
        h = C.c_void_p()
        _lib.check(lib.roma_create(C.byref(cfg), C.byref(h)), exc=AssertionError, lib=lib)
        try:
            for prefix, sd in (("", self._weights), ("dinov2.", self._dinov2_weights)):
                for k, v in sd.items():
                    t = v.detach().cpu().contiguous()
                    is_i64 = t.dtype == torch.int64
                    if not is_i64:
                        t = t.float().contiguous()
                    shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                    _lib.check(lib.roma_set_tensor(h, (prefix + k).encode(), t.dim(), shape, C.c_void_p(t.data_ptr()), int(is_i64)), lib=lib)
            # strict key/shape contract, as matcher.load_state_dict(weights) (roma_models.py:204)
            _lib.check(lib.roma_finalize(h), exc=RuntimeError, lib=lib)
        except Exception:
            lib.roma_destroy(h)
            raise
        return h

    def _release(self):
        for h in getattr(self, "_cache", {}).values():
            self._lib.roma_destroy(h)
        if getattr(self, "_cache", None) is not None:
            self._cache.clear()
        self._handle = self._built = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ nn.Module-ish surface used by callers
    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise _lib.RomaHipError("roma_amd runs only on a HIP device; there is no CPU fallback")
        return self

    def _get_device(self):
        return self.device

    def get_output_resolution(self):
        if not self.upsample_preds:
            return self.h_resized, self.w_resized
        return self.upsample_res

    # ------------------------------------------------------------------ match()
    @torch.inference_mode()
    def match(self, im_A_input, im_B_input, *args, im_A_high_res=None, im_B_high_res=None, batched=True, device=None):
        """romatch/models/matcher.py:779-934.  Returns (warp [B,H,2W,4] | [B,H,W,4], certainty [B,H,2W] | [B,H,W])."""
        from PIL import Image
        self.train(False)
        if not batched:
            raise ValueError("batched must be True, non-batched inference is no longer supported.")
        if device is None:
            device = im_A_input.device if isinstance(im_A_input, torch.Tensor) else self.device
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.RomaHipError(f"roma_amd.match needs CUDA/HIP tensors (got {device}); there is no CPU fallback")
        im_A = _check_input(im_A_input)
        im_B = _check_input(im_B_input)
        hs, ws = self.h_resized, self.w_resized
        scale_factor = math.sqrt(hs * ws / (560 ** 2))  # matcher.py:805: from the CONFIGURED resolution
        call_hw = None
        if isinstance(im_A, Image.Image) and isinstance(im_B, Image.Image):
            a = _pil_to_normalised(im_A, (hs, ws))[None].to(device)
            b_ = _pil_to_normalised(im_B, (hs, ws))[None].to(device)
        elif isinstance(im_A, torch.Tensor) and isinstance(im_B, torch.Tensor):
            b, c, h, w = im_A.shape
            b, c, h2, w2 = im_B.shape
            assert w == w2 and h == h2, "For batched images we assume same size"
            a, b_ = im_A.to(device), im_B.to(device)
            if h != self.h_resized or self.w_resized != w:
                warn("Model resolution and batch resolution differ, may produce unexpected results")
                call_hw = (h, w)  # the HIP handle is resolution-specific; the attributes stay as configured (matcher.py:822-826)
        else:
            raise ValueError(f"Unsupported input type: {type(im_A)=} and {type(im_B)=}")
        a_hr = b_hr = None
        if self.upsample_preds and im_A_high_res is None and im_B_high_res is None:
            assert isinstance(im_A, Image.Image), f"Unsupported input type: {type(im_A_input)=}"
            assert isinstance(im_B, Image.Image), f"Unsupported input type: {type(im_B_input)=}"
            a_hr = _pil_to_normalised(im_A, self.upsample_res)[None].to(device)
            b_hr = _pil_to_normalised(im_B, self.upsample_res)[None].to(device)
        elif self.upsample_preds and im_A_high_res is not None and im_B_high_res is not None:
            a_hr, b_hr = im_A_high_res.to(device), im_B_high_res.to(device)
            # The reference never touches self.upsample_res: it sizes the attenuation map and the output grid from the
            # attribute (matcher.py:836-838, 904-911) and the upsample pass from the tensors, so tensors of another size end
            # in a RuntimeError (shape mismatch at matcher.py:891-894).  Same exception here, raised before any work.
            if tuple(a_hr.shape[-2:]) != tuple(self.upsample_res) or tuple(b_hr.shape[-2:]) != tuple(self.upsample_res):
                raise RuntimeError(f"im_A_high_res / im_B_high_res are {tuple(a_hr.shape[-2:])} / {tuple(b_hr.shape[-2:])} "
                                   f"but upsample_res is {tuple(self.upsample_res)}: the size of the high-resolution "
                                   "tensors must match upsample_res (set the attribute before calling match())")
        elif self.upsample_preds:
            raise ValueError(f"Invalid upsample_preds and high_res inputs with {im_A_high_res=} and {im_B_high_res=}")
        if device.index is not None and device.index != self.device.index:
            raise ValueError(f"roma_amd.match: inputs live on {device} but this matcher was built for {self.device}; "
                             "use one matcher (one handle) per GPU")
        self._ensure_handle(call_hw)
        lib = self._lib
        for k in ("symmetric", "upsample_preds", "attenuate_cert", "debug", "vit_bf16_residual", "dual_stream", "trace", "compose_out_conv"):
            _lib.check(lib.roma_set_option(self._handle, k.encode(), int(bool(getattr(self, k)))), lib=lib)
        _lib.check(lib.roma_set_option_f(self._handle, b"coarse_scale_factor", float(scale_factor)), lib=lib)
        B = a.shape[0]
        Ho, Wo = self.get_output_resolution() if self.upsample_preds else (a.shape[-2], a.shape[-1])
        Wout = 2 * Wo if self.symmetric else Wo
        warp = torch.empty((B, Ho, Wout, 4), device=device, dtype=torch.float32)
        cert = torch.empty((B, Ho, Wout), device=device, dtype=torch.float32)
        a, b_ = a.float().contiguous(), b_.float().contiguous()
        if a_hr is not None:
            a_hr, b_hr = a_hr.float().contiguous(), b_hr.float().contiguous()
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            for i0 in range(0, B, self.max_batch):
                n = min(self.max_batch, B - i0)
                ptr = lambda t: C.c_void_p(t[i0:i0 + n].data_ptr()) if t is not None else None  # noqa: E731
                _lib.check(lib.roma_match(self._handle, n, ptr(a), ptr(b_), ptr(a_hr), ptr(b_hr),
                                          ptr(warp), ptr(cert), C.c_void_p(stream)))
        return warp, cert

    # ------------------------------------------------------------------ forward / forward_symmetric / backbone features
    _SCALES = (16, 8, 4, 2, 1)
    _FEAT_C = {16: 1024, 8: 512, 4: 256, 2: 128, 1: 64}

    def _scale_hw(self, H, W, s):
        return (H // 14, W // 14) if s == 16 else (H // s, W // s)

    def _act_torch_dtype(self):
        if self.amp_dtype == torch.float32:
            return torch.float32
        return torch.float16 if self._lib.h16 == "f16" else torch.bfloat16

    @torch.inference_mode()
    def _forward_pass(self, batch, symmetric, upsample, scale_factor, want_corresps=True, want_feats=False):
        """One roma_forward call: the decoder pass of matcher.py:631-670 (and / or the feature pyramid of :585-596)."""
        im_A, im_B = batch["im_A"], batch["im_B"]
        if not (isinstance(im_A, torch.Tensor) and isinstance(im_B, torch.Tensor)) or not im_A.is_cuda:
            raise _lib.RomaHipError("roma_amd.forward needs CUDA/HIP tensors; there is no CPU fallback")
        b, c, H, W = im_A.shape
        assert tuple(im_B.shape) == tuple(im_A.shape), "For batched images we assume same size"
        dev = im_A.device
        # same device contract as match(): the handle does hipSetDevice(self.device), so a tensor of another GPU (or a CPU
        # im_B) would reach the kernels as a foreign pointer - a memory fault instead of a Python exception
        if dev.index is not None and dev.index != self.device.index:
            raise ValueError(f"roma_amd.forward: inputs live on {dev} but this matcher was built for {self.device}; "
                             "use one matcher (one handle) per GPU")
        dev = self.device
        im_A, im_B = im_A.to(dev), im_B.to(dev)
        if upsample:
            if (H, W) != tuple(self.upsample_res):
                raise RuntimeError(f"forward(upsample=True): images are {(H, W)} but upsample_res is {tuple(self.upsample_res)}")
            keep = self.upsample_preds
            self.upsample_preds = True  # the handle must be planned for the upsample pass
            try:
                self._ensure_handle()
            finally:
                self.upsample_preds = keep
        else:
            self._ensure_handle((H, W) if (H, W) != (self.h_resized, self.w_resized) else None)
        if b > self.max_batch:
            raise ValueError(f"forward: batch {b} > max_batch {self.max_batch}")
        lib = self._lib
        for k in ("debug", "vit_bf16_residual", "trace", "compose_out_conv"):
            _lib.check(lib.roma_set_option(self._handle, k.encode(), int(bool(getattr(self, k)))), lib=lib)
        bd = 2 * b if symmetric else b
        fa = _lib.RomaForwardArgs()
        fa.upsample, fa.symmetric, fa.scale_factor = int(bool(upsample)), int(bool(symmetric)), float(scale_factor)
        keepalive = []
        if upsample:
            cor = batch.get("corresps")
            if cor is None and want_corresps:
                raise ValueError("forward(upsample=True) needs batch['corresps'] = {'flow', 'certainty'} (matcher.py:653-657)")
            if cor is None:  # feature pyramid only: the decoder pass still runs (one schedule), on a zero seed
                cor = {"flow": torch.zeros((bd, 2, 1, 1), device=dev), "certainty": torch.zeros((bd, 1, 1, 1), device=dev)}
            sf = cor["flow"].to(dev, torch.float32).permute(0, 2, 3, 1).contiguous()        # [Bd, h, w, 2]
            sc = cor["certainty"].to(dev, torch.float32).reshape(bd, *cor["certainty"].shape[-2:]).contiguous()
            assert sf.shape[0] == bd and tuple(sc.shape[-2:]) == tuple(sf.shape[1:3])
            fa.seed_flow, fa.seed_cert, fa.seed_h, fa.seed_w = sf.data_ptr(), sc.data_ptr(), sf.shape[1], sf.shape[2]
            keepalive += [sf, sc]
        flows, certs, feats = {}, {}, {}
        for i, s in enumerate(self._SCALES):
            if upsample and s == 16:
                continue
            hs, ws = self._scale_hw(H, W, s)
            if want_corresps:
                flows[s] = torch.empty((bd, hs, ws, 2), device=dev, dtype=torch.float32)
                certs[s] = torch.empty((bd, hs, ws), device=dev, dtype=torch.float32)
                fa.flow[i], fa.cert[i] = flows[s].data_ptr(), certs[s].data_ptr()
            if want_feats:
                feats[s] = torch.empty((2 * b, hs, ws, self._FEAT_C[s]), device=dev, dtype=self._act_torch_dtype())
                fa.feat[i] = feats[s].data_ptr()
        a, b_ = im_A.float().contiguous(), im_B.float().contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(lib.roma_forward(self._handle, b, C.c_void_p(a.data_ptr()), C.c_void_p(b_.data_ptr()), C.byref(fa),
                                        C.c_void_p(stream)), lib=lib)
        corresps = {s: {"certainty": certs[s][:, None], "flow": flows[s].permute(0, 3, 1, 2)} for s in flows}
        pyramid = {s: f.permute(0, 3, 1, 2) for s, f in feats.items()}
        return corresps, pyramid

    def extract_backbone_features(self, batch, batched=True, upsample=False):
        """matcher.py:585-596: {scale: [2B, C, h, w]} (A images then B images; NCHW views of the channels-last device
        tensors, in the handle's activation dtype).  upsample=True returns the VGG scales only (encoders.py:56-68)."""
        return self._forward_pass(batch, symmetric=False, upsample=upsample, scale_factor=1.0, want_corresps=False,
                                  want_feats=True)[1]

    def forward(self, batch, batched=True, upsample=False, scale_factor=1):
        """matcher.py:631-651 in eval mode: corresps[s] = {"certainty" [B,1,h,w] logits, "flow" [B,2,h,w]} for A -> B."""
        return self._forward_pass(batch, symmetric=False, upsample=upsample, scale_factor=scale_factor)[0]

    def forward_symmetric(self, batch, batched=True, upsample=False, scale_factor=1):
        """matcher.py:653-670: the same with the decoder batch doubled - entries [0, B) are A -> B, [B, 2B) are B -> A."""
        return self._forward_pass(batch, symmetric=True, upsample=upsample, scale_factor=scale_factor)[0]

    def debug_fetch(self, name: str, dtype=np.float32) -> np.ndarray:
        """Intermediate tensor captured by the last match() when `self.debug` is set (tests only)."""
        lib = self._lib
        n = lib.roma_debug_fetch(self._handle, name.encode(), None, 0)
        if n < 0:
            raise KeyError(_lib.last_error(lib))
        buf = np.empty(n // np.dtype(dtype).itemsize, dtype=dtype)
        got = lib.roma_debug_fetch(self._handle, name.encode(), C.c_void_p(buf.ctypes.data), n)
        if got < 0:
            raise _lib.RomaHipError(_lib.last_error(lib))
        return buf

    def debug_trace(self, slot: int = 0):
        """(names, uint64 checksums) of the stages of the last match() on sub-batch stream `slot` (needs `self.trace`)."""
        lib = self._lib
        n = lib.roma_debug_trace(self._handle, slot, None, 0, None, 0)
        if n <= 0:
            return [], np.zeros(0, dtype=np.uint64)
        sums = np.zeros(n, dtype=np.uint64)
        names = C.create_string_buffer(64 * n + 16)
        got = lib.roma_debug_trace(self._handle, slot, C.c_void_p(sums.ctypes.data), n, names, len(names))
        if got < 0:
            raise _lib.RomaHipError(_lib.last_error(lib))
        return names.value.decode().split("\n")[:n], sums

    def debug_inject(self, name: str, value: Optional[np.ndarray]):
        """Tests only (needs `self.debug`): override the named intermediate ("gm_flow16" [b,T,2], "gm_cert16" [b,T]) of
        the following match() calls with `value`; None removes the override."""
        lib = self._lib
        self._ensure_handle()
        if value is None:
            _lib.check(lib.roma_debug_inject(self._handle, name.encode(), None, 0), lib=lib)
            return
        a = np.ascontiguousarray(value, dtype=np.float32)
        _lib.check(lib.roma_debug_inject(self._handle, name.encode(), C.c_void_p(a.ctypes.data), a.nbytes), lib=lib)

    # ------------------------------------------------------------------ sampling (matcher.py:598-629)
    def sample(self, matches, certainty, num=10000):
        """Certainty-weighted sampling of matches, optionally balanced by the inverse match density.

        Same control flow and attributes (`sample_mode`, `sample_thresh`) as the reference; the density is
        `roma_amd.kde.kde` (HIP) and the two draws without replacement are `roma_amd.sampling.multinomial` (HIP exponential
        race + radix select).  The result is stochastic, so parity with the reference is distributional."""
        from .kde import kde
        from .sampling import multinomial
        if "threshold" in self.sample_mode:
            upper_thresh = self.sample_thresh
            certainty = certainty.clone()
            certainty[certainty > upper_thresh] = 1
        matches, certainty = matches.reshape(-1, 4), certainty.reshape(-1)
        expansion_factor = 4 if "balanced" in self.sample_mode else 1
        good_samples = multinomial(certainty, min(expansion_factor * num, len(certainty)))
        good_matches, good_certainty = matches[good_samples], certainty[good_samples]
        if "balanced" not in self.sample_mode:
            return good_matches, good_certainty
        density = kde(good_matches, std=0.1)
        p = 1 / (density + 1)
        p[density < 10] = 1e-7  # at least 10 perfect neighbours, or around 100 ok ones (matcher.py:622-624)
        balanced_samples = multinomial(p, min(num, len(good_certainty)))
        return good_matches[balanced_samples], good_certainty[balanced_samples]

    # ------------------------------------------------------------------ keypoint matching (matcher.py:732-773)
    def match_keypoints(self, x_A, x_B, warp, certainty, return_tuple=True, return_inds=False, max_dist=0.005, cert_th=0):
        """Mutual-nearest-neighbour matching of detector keypoints through the dense warp.

        x_A [Na,2], x_B [Nb,2] normalised (x,y); warp [H,W,4] / certainty [H,W] of ONE pair (as returned by match() with
        the batch dimension removed).  `roma_op_sample_warp_at` + `roma_op_mutual_nn_count / _fill` replace the reference's
        grid_sample + Na x Nb cdist matrix and return what its torch.nonzero returns: every mutual pair, tied pairs
        (duplicate keypoints) included, in row-major order."""
        for t in (x_A, x_B, warp, certainty):
            if not t.is_cuda:
                raise _lib.RomaHipError("match_keypoints: tensors must live on a HIP device; there is no CPU fallback")
        lib = _lib.load()
        H, W = int(warp.shape[0]), int(warp.shape[1])
        if warp.dim() != 3 or warp.shape[2] != 4 or tuple(certainty.shape) != (H, W):
            raise ValueError("match_keypoints: expected warp [H,W,4] and certainty [H,W]")
        w = warp.detach().to(torch.float32).contiguous()
        c = certainty.detach().to(torch.float32).contiguous()
        xa = x_A.detach().to(torch.float32).contiguous()
        xb = x_B.detach().to(torch.float32).contiguous()
        na, nb = xa.shape[0], xb.shape[0]
        dev = xa.device
        xab = torch.empty((na, 2), device=dev, dtype=torch.float32)
        ca = torch.empty((na,), device=dev, dtype=torch.float32)
        offs = torch.empty((na + 1,), device=dev, dtype=torch.int64)
        ws_a = torch.empty((max(na, 1),), device=dev, dtype=torch.int64)
        ws_b = torch.empty((max(nb, 1),), device=dev, dtype=torch.int64)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(lib.roma_op_sample_warp_at(P(w), P(c), H, W, P(xa), na, P(xab), P(ca), stream))
            _lib.check(lib.roma_op_mutual_nn_count(P(xab), na, P(xb), nb, P(ca), float(cert_th), float(max_dist),
                                                   P(ws_a), P(ws_b), P(offs), stream))
            n_pairs = int(offs[na].item())  # the one host read (torch.nonzero synchronises in the reference too)
            pairs = torch.empty((n_pairs, 2), device=dev, dtype=torch.int64)
            if n_pairs:
                _lib.check(lib.roma_op_mutual_nn_fill(P(xab), na, P(xb), nb, P(ca), float(cert_th), float(max_dist),
                                                      P(ws_a), P(ws_b), P(offs), P(pairs), stream))
        inds_A, inds_B = pairs[:, 0], pairs[:, 1]
        if return_tuple:
            if return_inds:
                return inds_A, inds_B
            return x_A[inds_A], x_B[inds_B]
        if return_inds:
            return torch.cat((inds_A, inds_B), dim=-1)
        return torch.cat((x_A[inds_A], x_B[inds_B]), dim=-1)

    def conf_from_fb_consistency(self, flow_forward, flow_backward, th=2):
        """matcher.py:672-699: 1 where the backward flow maps the forward target back to within `th` pixels."""
        for t in (flow_forward, flow_backward):
            if not t.is_cuda:
                raise _lib.RomaHipError("conf_from_fb_consistency: tensors must live on a HIP device; there is no CPU fallback")
        has_batch = flow_forward.dim() != 3
        ff = (flow_forward if has_batch else flow_forward[None]).detach().to(torch.float32).contiguous()
        fb = (flow_backward if has_batch else flow_backward[None]).detach().to(torch.float32).contiguous()
        B, H, W = int(ff.shape[0]), int(ff.shape[-3]), int(ff.shape[-2])
        out = torch.empty((B, H, W), device=ff.device, dtype=torch.float32)
        with torch.cuda.device(ff.device):
            _lib.check(_lib.load().roma_op_fb_consistency(C.c_void_p(ff.data_ptr()), C.c_void_p(fb.data_ptr()), B, H, W,
                                                          float(2 * th / max(H, W)), C.c_void_p(out.data_ptr()),
                                                          C.c_void_p(torch.cuda.current_stream(ff.device).cuda_stream)))
        return out if has_batch else out[0]

    def visualize_warp(self, warp, certainty, im_A=None, im_B=None, im_A_path=None, im_B_path=None, device="cuda",
                       symmetric=True, save_path=None, unnormalize=False):
        """matcher.py:936-986: both images warped into each other and blended with the certainty over a white background
        (one HIP kernel: bilinear grid_sample + blend).  Same arguments and return value ([3, H, W2] tensor) as the
        reference; PIL / path inputs are resized to the warp's resolution on the host like there."""
        from PIL import Image
        dev = warp.device
        if not warp.is_cuda:
            raise _lib.RomaHipError("visualize_warp: warp / certainty must live on a HIP device; there is no CPU fallback")
        H, W2, _ = warp.shape
        W = W2 // 2 if symmetric else W2
        if im_A is None:
            im_A, im_B = Image.open(im_A_path).convert("RGB"), Image.open(im_B_path).convert("RGB")
        if not isinstance(im_A, torch.Tensor):
            im_A, im_B = im_A.resize((W, H)), im_B.resize((W, H))
            x_B = (torch.tensor(np.array(im_B)) / 255).to(dev).permute(2, 0, 1)
            x_A = (torch.tensor(np.array(im_A)) / 255).to(dev).permute(2, 0, 1) if symmetric else None
        else:
            x_A, x_B = (im_A if symmetric else None), im_B
        x_B = x_B.detach().to(dev, torch.float32).contiguous()
        x_A = x_A.detach().to(dev, torch.float32).contiguous() if x_A is not None else None
        if x_A is not None and tuple(x_A.shape) != tuple(x_B.shape):
            raise ValueError("visualize_warp: im_A and im_B must have the same shape")
        w = warp.detach().to(torch.float32).contiguous()
        c = certainty.detach().to(torch.float32).contiguous()
        out = torch.empty((3, H, W2), device=dev, dtype=torch.float32)
        P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(_lib.load().roma_op_visualize_warp(P(w), P(c), P(x_A), P(x_B), H, W, int(bool(symmetric)),
                                                          int(x_B.shape[-2]), int(x_B.shape[-1]), P(out),
                                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        if save_path is not None:
            vis = out
            if unnormalize:  # utils.tensor_to_pil(unnormalize=True): undo the ImageNet normalisation
                mean = torch.tensor([0.485, 0.456, 0.406], device=dev)[:, None, None]
                std = torch.tensor([0.229, 0.224, 0.225], device=dev)[:, None, None]
                vis = vis * std + mean
            arr = (vis.permute(1, 2, 0).clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)
            Image.fromarray(arr).save(save_path)
        return out

    # ------------------------------------------------------------------ light post-processing helpers (torch)
    def to_pixel_coordinates(self, coords, H_A, W_A, H_B=None, W_B=None):
        """matcher.py:701-717."""
        if coords.shape[-1] == 2:
            return self._to_pixel_coordinates(coords, H_A, W_A)
        if isinstance(coords, (list, tuple)):
            kpts_A, kpts_B = coords[0], coords[1]
        else:
            kpts_A, kpts_B = coords[..., :2], coords[..., 2:]
        return self._to_pixel_coordinates(kpts_A, H_A, W_A), self._to_pixel_coordinates(kpts_B, H_B, W_B)

    @staticmethod
    def _to_pixel_coordinates(coords, H, W):
        return torch.stack((W / 2 * (coords[..., 0] + 1), H / 2 * (coords[..., 1] + 1)), dim=-1)

    def to_normalized_coordinates(self, coords, H_A, W_A, H_B, W_B):
        """matcher.py:719-730."""
        if isinstance(coords, (list, tuple)):
            kpts_A, kpts_B = coords[0], coords[1]
        else:
            kpts_A, kpts_B = coords[..., :2], coords[..., 2:]
        kpts_A = torch.stack((2 / W_A * kpts_A[..., 0] - 1, 2 / H_A * kpts_A[..., 1] - 1), dim=-1)
        kpts_B = torch.stack((2 / W_B * kpts_B[..., 0] - 1, 2 / H_B * kpts_B[..., 1] - 1), dim=-1)
        return kpts_A, kpts_B


def roma_model(resolution, upsample_preds, device=None, weights=None, dinov2_weights=None,
               amp_dtype: torch.dtype = torch.float16, use_custom_corr=True, symmetric=True, upsample_res=None,
               sample_thresh=0.05, sample_mode="threshold_balanced", attenuate_cert=True, max_batch=8, **kwargs):
    """romatch/models/model_zoo/roma_models.py:32-205.  `use_custom_corr` is accepted for API
    compatibility; the fused HIP local-correlation kernel is always used.

    Accepted: `amp_dtype` float32 (exact-f32 MFMA parity mode), bfloat16 (libroma_hip.so), float16 (libroma_hip_f16.so);
    `resolution` a multiple of 14 per side (the DINOv2 patch grid, asserted as in the reference); sides that are not
    multiples of 8 are fine - the VGG pyramid floors at every max-pool like the reference's (126 x 154 -> 182 x 198 is a
    tested configuration); `upsample_res` any size the reference accepts."""
    resolution = _to_hw(resolution)
    upsample_res = _to_hw(upsample_res)
    assert resolution[0] % 14 == 0, "Needs to be multiple of 14 for backbone"
    assert resolution[1] % 14 == 0, "Needs to be multiple of 14 for backbone"
    if weights is None or dinov2_weights is None:
        raise ValueError("weights and dinov2_weights state-dicts are required (no network access for torch.hub downloads)")
    h, w = resolution
    return RegressionMatcher(weights, dinov2_weights, h=h, w=w, upsample_preds=upsample_preds, upsample_res=upsample_res,
                             symmetric=symmetric, attenuate_cert=attenuate_cert, sample_mode=sample_mode,
                             sample_thresh=sample_thresh, device=device, amp_dtype=amp_dtype, max_batch=max_batch, **kwargs)


def roma_outdoor(device, weights=None, dinov2_weights=None, coarse_res: Union[int, tuple] = 560,
                 upsample_res: Union[int, tuple] = 864, amp_dtype: torch.dtype = torch.float16, symmetric=True,
                 use_custom_corr=True, upsample_preds=True, max_batch=8, decoder_dtype=None):
    """romatch/models/model_zoo/__init__.py:31-61.  `decoder_dtype=torch.float16` with `amp_dtype=torch.bfloat16` = the
    precision mix of the reference's timing script (RegressionMatcher.__init__)."""
    return roma_model(resolution=coarse_res, upsample_preds=upsample_preds, weights=weights,
                      dinov2_weights=dinov2_weights, device=device, amp_dtype=amp_dtype, symmetric=symmetric,
                      use_custom_corr=use_custom_corr, upsample_res=upsample_res, max_batch=max_batch,
                      decoder_dtype=decoder_dtype)


def roma_indoor(device, weights=None, dinov2_weights=None, coarse_res: Union[int, tuple] = 560,
                upsample_res: Union[int, tuple] = 864, amp_dtype: torch.dtype = torch.float16, symmetric=True,
                use_custom_corr=True, upsample_preds=True, max_batch=8, decoder_dtype=None):
    """romatch/models/model_zoo/__init__.py:64-93 (same graph as roma_outdoor, different weights)."""
    return roma_model(resolution=coarse_res, upsample_preds=upsample_preds, weights=weights,
                      dinov2_weights=dinov2_weights, device=device, amp_dtype=amp_dtype, symmetric=symmetric,
                      use_custom_corr=use_custom_corr, upsample_res=upsample_res, max_batch=max_batch,
                      decoder_dtype=decoder_dtype)
