// Device side of Tiny RoMa's matcher (romatch/models/tiny.py:114-142, 182-196, 222-238, 278-303); see tiny.hip.
#pragma once
#include "common.h"

namespace roma {

int nchw_to_nhwc_launch(const float* in, float* out, int B, int C, int H, int W, hipStream_t s);
int tiny_pos_embed_launch(const float* cv, float* out, int B, int H1, int W1, int H0, int W0, int exact_softmax, hipStream_t s);
// XFeat-style backbone layers (tiny.py:81-99), channels-last f32
int gray_instnorm_launch(const float* in, float* out, int B, int H, int W, int C, float eps, hipStream_t s);
int conv2d_nhwc_launch(const float* in, const float* w, const float* bias, const float* res, float* out, int B, int H, int W,
                       int Cin, int Cout, int K, int stride, int pad, int relu, hipStream_t s);
int avgpool_nhwc_launch(const float* in, float* out, int B, int H, int W, int C, int k, hipStream_t s);
int add3_launch(const float* a, const float* b, const float* c, float* out, long n, hipStream_t s);
int tiny_matcher_input_launch(const float* f0, const float* f1, const float* warp, int warp_channels, float* d, int B, int H,
                              int W, int H1, int W1, int C, int Cp, hipStream_t s);
int tiny_update_launch(const float* base, int base_channels, const float* delta, long ldd, float sx, float sy, float* out,
                       long npix, hipStream_t s);
int tiny_final_launch(const float* matches, float* warp, float* cert, int B, int H, int W, hipStream_t s);

}  // namespace roma
