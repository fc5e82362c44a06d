// Weight-stationary 3x3 convolutions for Cin = 64 (VGG conv1_2 / conv2_1) and Cin = Cout = 128 (conv2_2): see conv64.hip.
#pragma once
#include "gemm.h"

namespace roma {
// 0 = launched, 1 = not this kernel's problem (the implicit GEMM runs it), < 0 = error.  `a` as built for gemm_launch.
int conv64_try_launch(const GemmArgs& a, hipStream_t stream);
extern int g_conv64_mode;  // roma_tuning("conv64", v): bit 0 the Cin = 64 kernels, bit 1 the Cin = 128 kernel; -1 = env ROMA_CONV64 (default 3)
}  // namespace roma
