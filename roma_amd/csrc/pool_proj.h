// MaxPool 2x2 + proj head of a VGG pyramid level in one pass (pool_proj.hip); 16-bit maps, C = 64 / 128.
#pragma once
#include "common.h"

namespace roma {

// in [nimg, H, W, C] -> pooled [nimg, H/2, W/2, C] (floor, as nn.MaxPool2d) and pf [nimg, H*W, ldf] = in . pw^T + pb
// (pw [N][ldw] 16-bit, pb f32 [N]; columns N .. ldf of pf are written as zeros).  Bit-identical to maxpool2x2_launch + the proj GEMM.
bool pool_proj_supported(int C, int N, int ldf, int dt);
int pool_proj_launch(const void* in, void* pooled, void* pf, const void* pw, long ldw, const float* pb, int N, int ldf, int nimg,
                     int H, int W, int C, int dt, hipStream_t s);

}  // namespace roma
