// Weight-stationary 3x3 convolutions for Cin = 64 (VGG conv1_2 / conv2_1) and Cin = Cout = 128 (conv2_2): see conv64.hip.
#pragma once
#include "gemm.h"

namespace roma {
// 0 = launched, 1 = not this kernel's problem (the implicit GEMM runs it), < 0 = error.  `a` as built for gemm_launch.
int conv64_try_launch(const GemmArgs& a, hipStream_t stream);
// First VGG layer in bf16 mode, fused: f32 NCHW image [B,3,H,W] -> bf16 NHWC [B,H,W,64] = ReLU(conv3x3(img, w) + bias).
// w: bf16 [64][32], k = ci * 9 + ky * 3 + kx, columns 27..31 zero (Model::vgg[0]).
int conv3x3_c3_bf16_launch(const float* img, const void* w, const float* bias, void* out, int B, int H, int W, hipStream_t stream);
extern int g_conv64_mode;  // roma_tuning("conv64", v): bit 0 the Cin = 64 kernels, bit 1 the Cin = 128 kernel, bit 2 the fused first layer (model.hip); -1 = env ROMA_CONV64 (default 7)
}  // namespace roma
