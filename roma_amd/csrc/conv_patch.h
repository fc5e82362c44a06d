// Patch-resident 3x3 convolution for the VGG layers with Cout >= 256 (conv_patch.hip).
#pragma once
#include "gemm.h"

namespace roma {

// Takes the implicit-GEMM description of gemm.h (conv_c > 0, conv_korder = 1: slab-major weight rows, 16-bit in / out, bias +
// ReLU, Cin in {128 .. 512} % 64 == 0, Cout % 256 == 0).  0 = launched, 1 = not this kernel's problem, < 0 = error.
bool conv_patch_supported(const GemmArgs& a);
int conv_patch_try_launch(const GemmArgs& a, hipStream_t stream);
extern int g_conv_patch;  // roma_tuning("conv_patch")

}  // namespace roma
