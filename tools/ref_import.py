"""Import the UNMODIFIED reference (`/root/reference`, Parskatt/RoMa) in this build
container for golden-vector generation.  Build-container only: `/root/reference`
does not exist on the GPU box, so nothing under tests/ (gpu marker), bench.py or
smoke() may import this module.

The image lacks cv2 / loguru / torchvision, which the reference imports at module
top (romatch/utils/utils.py:3,6-7, romatch/models/encoders.py:3,
romatch/models/model_zoo/roma_models.py:7).  We register stub modules that
contribute NO arithmetic to the tensor route except the VGG19-BN *layer list* (cfg "E" of
torchvision: Conv3x3(pad 1)+BatchNorm2d+ReLU / MaxPool2d(2,2)), which the reference slices as
`features[:40]` (encoders.py:13).  For the path / PIL route the three torchvision transforms
the reference calls (Resize on a PIL image, ToTensor, Normalize) are restated from
torchvision's published definitions (round 5; see install_stubs).  Everything else is the
reference's own code.
"""
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def _vgg19_bn_features():
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
           512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    layers, c_in = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c_in, v, kernel_size=3, padding=1),
                       nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            c_in = v
    return nn.Sequential(*layers)


def install_stubs():
    if "romatch" in sys.modules:
        return
    cv2 = types.ModuleType("cv2")
    loguru = types.ModuleType("loguru")

    class _Logger:
        def info(self, *a, **k):
            pass
        warning = debug = error = info

    loguru.logger = _Logger()

    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm_vgg = types.ModuleType("torchvision.models.vgg")
    tvt = types.ModuleType("torchvision.transforms")
    tvtf = types.ModuleType("torchvision.transforms.functional")

    class _VGG:
        def __init__(self):
            self.features = _vgg19_bn_features()

    def vgg19_bn(weights=None):
        assert weights is None, "no pretrained weights offline"
        return _VGG()

    class VGG19_BN_Weights:
        IMAGENET1K_V1 = None

    tvm.vgg19_bn = vgg19_bn
    tvm.vgg = tvm_vgg
    tvm_vgg.VGG19_BN_Weights = VGG19_BN_Weights

    class InterpolationMode:
        BICUBIC = "bicubic"
        BILINEAR = "bilinear"

    tvtf.InterpolationMode = InterpolationMode

    # Path / PIL inputs (matcher.py:806-816, 853-868; tiny.py:193-198) go through three torchvision transforms.  torchvision
    # is absent, so the stubs restate what the pinned torchvision does for a PIL input - nothing of the reference itself:
    #   Resize(size, BICUBIC)(pil)  = pil.resize(size[::-1], PIL.Image.BICUBIC)      (transforms/_functional_pil.py resize)
    #   ToTensor()(pil)             = uint8 HWC -> CHW, .to(float32).div(255)        (transforms/functional.py to_tensor)
    #   Normalize(mean, std)(t)     = (t - mean[:, None, None]) / std[:, None, None] (functional.py normalize, float32)
    class Resize:
        def __init__(self, size, interpolation=InterpolationMode.BILINEAR, *a, **k):
            self.size = (size, size) if isinstance(size, int) else tuple(size)
            self.interpolation = interpolation

        def __call__(self, im):
            from PIL import Image
            assert isinstance(im, Image.Image), "Resize stub: PIL input only (the path / PIL route of match())"
            mode = {InterpolationMode.BICUBIC: Image.BICUBIC, InterpolationMode.BILINEAR: Image.BILINEAR}[self.interpolation]
            return im.resize((self.size[1], self.size[0]), mode)

    class Normalize:
        def __init__(self, mean, std, inplace=False):
            self.mean, self.std = mean, std

        def __call__(self, t):
            mean = torch.as_tensor(self.mean, dtype=t.dtype)[:, None, None]
            std = torch.as_tensor(self.std, dtype=t.dtype)[:, None, None]
            return (t.clone() - mean) / std

    class ToTensor:
        def __call__(self, im):
            import numpy as np
            a = torch.from_numpy(np.array(im, dtype=np.uint8, copy=True))
            if a.dim() == 2:
                a = a[:, :, None]
            return a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tvt.Resize, tvt.Normalize, tvt.ToTensor, tvt.Compose = Resize, Normalize, ToTensor, Compose
    tvt.functional = tvtf
    tv.models = tvm
    tv.transforms = tvt
    for name, mod in [("cv2", cv2), ("loguru", loguru), ("torchvision", tv),
                      ("torchvision.models", tvm), ("torchvision.models.vgg", tvm_vgg),
                      ("torchvision.transforms", tvt), ("torchvision.transforms.functional", tvtf)]:
        sys.modules.setdefault(name, mod)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def build_reference_matcher(weights, dinov2_weights, coarse_res, upsample_res,
                            symmetric=True, upsample_preds=True, attenuate_cert=True):
    """roma_model(...) of the reference on CPU (fp32; roma_models.py:55-56), torch
    local-corr fallback (local_correlation.py:39-74)."""
    install_stubs()
    from romatch.models.model_zoo.roma_models import roma_model
    m = roma_model(resolution=coarse_res, upsample_preds=upsample_preds, device="cpu",
                   weights=weights, dinov2_weights=dinov2_weights, upsample_res=upsample_res,
                   use_custom_corr=False, symmetric=symmetric, attenuate_cert=attenuate_cert)
    return m.eval()
