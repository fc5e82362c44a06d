// Device side of Tiny RoMa's matcher (romatch/models/tiny.py:114-142, 182-196, 222-238, 278-303); see tiny.hip.
#pragma once
#include "common.h"

namespace roma {

int nchw_to_nhwc_launch(const float* in, float* out, int B, int C, int H, int W, hipStream_t s);
int tiny_pos_embed_launch(const float* cv, float* out, int B, int H1, int W1, int H0, int W0, hipStream_t s);
int tiny_matcher_input_launch(const float* f0, const float* f1, const float* warp, int warp_channels, float* d, int B, int H,
                              int W, int H1, int W1, int C, int Cp, hipStream_t s);
int tiny_update_launch(const float* base, int base_channels, const float* delta, long ldd, float sx, float sy, float* out,
                       long npix, hipStream_t s);
int tiny_final_launch(const float* matches, float* warp, float* cert, int B, int H, int W, hipStream_t s);

}  // namespace roma
