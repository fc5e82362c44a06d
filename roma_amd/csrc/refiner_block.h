// Fused ConvRefiner block for the narrow fine scales (reference: romatch/models/matcher.py:88-117 create_block):
//   out = conv1x1( relu( bn( dwconv5x5(in) ) ) )      BN folded into the depthwise weights at pack time.
// The separate dwconv + GEMM pair moves the activation through HBM twice per block; at the fine scales (C = 144 at
// stride 2, C = 24 at stride 1) both kernels are purely bandwidth bound, so this kernel keeps the depthwise result
// in LDS, runs the 1x1 on MFMA out of LDS and writes the block output once (2 x instead of 4 x the tensor bytes).
// bf16 activations only (throughput mode); the exact-f32 mode keeps the two-kernel path.
#pragma once
#include "common.h"

namespace roma {
// in/out: [B,H,W,Cp] bf16 channels-last (must not alias); dw_w f32 [25][Cp], dw_b f32 [Cp] (BN folded);
// pw bf16 [Cp][ldpw] (K contiguous), pw_b f32 [Cp].  Supported Cp: 24, 144.  Returns 0 / negative error code.
bool refiner_block_supported(int Cp, int dt);
int refiner_block_launch(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw, long ldpw,
                         const float* pw_b, int B, int H, int W, int Cp, int dt, hipStream_t s);
// C = 24 only: the wave-private form (refiner_block24w.hip); 0 = launched, 1 = not taken (caller uses the workgroup kernel)
int refiner_block24_wave_try_launch(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw, long ldpw,
                                    const float* pw_b, int B, int H, int W, int dt, hipStream_t s, float* delta = nullptr);
// The LAST block of a narrow ConvRefiner (Cp = 24 / 144) with its 1x1 composed with out_conv (matcher.py:92-122, 175-178: two
// linear maps back to back): writes delta[pixel] = {d flow x, d flow y, d certainty, 0} instead of a block output.
// pw_final: 16-bit [8][ldpw], rows 0-2 = head, rows 4-6 = 16-bit remainder of the composed [3][Cp] weights; bias_final f32 [Cp].
int refiner_block_final_launch(const void* in, float* delta, const float* dw_w, const float* dw_b, const void* pw_final, long ldpw,
                               const float* bias_final, int B, int H, int W, int Cp, int dt, hipStream_t s);
// C = 576 (the stride-4 ConvRefiner, both passes): the whole block in one kernel with all 576 output channels per workgroup
// (refiner_block_wide.hip).  0 = launched, 1 = not taken (caller runs dwconv5x5 + 1x1 GEMM), < 0 = error.
bool refiner_block_wide_supported(int Cp, int dt);
int refiner_block_wide_try_launch(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw, long ldpw,
                                  const float* pw_b, int B, int H, int W, int Cp, int dt, hipStream_t s, bool force = false);
extern int g_rb_wide;
extern int g_rb24_wave;
extern int g_rb144_1b;
}  // namespace roma
