// Tiny RoMa (romatch/models/tiny.py): the matcher side of TinyRoMa.forward / match on the device.  The XFeat backbone is
// an un-vendored hub dependency of the reference (model_zoo/__init__.py:24-27); the caller supplies it and hands over its
// two feature maps.  Everything after forward_single (tiny.py:278-303, 222-242) runs here, channels-last, f32:
//   corr_volume (tiny.py:182-196)        -> the batched MFMA GEMM of gemm.hip (roma_op_gemm, alpha = 1/sqrt(C))
//   pos_embed, eval path (tiny.py:114-142) -> tiny_pos_embed_kernel
//   cat(f0, grid_sample(f1, warp), warp) (tiny.py:290-291, 298-299) -> tiny_matcher_input_kernel
//   BasicLayer x 4 + 1x1 (tiny.py:49-62)  -> the implicit-GEMM 3x3 convolution with folded BN + roma_op_gemm
//   matches += delta * to_normalized (tiny.py:292, 300) -> tiny_update_kernel
//   F.interpolate(bilinear) (tiny.py:294, 222-236) -> resize_bilinear_kernel (elementwise.hip)
//   warp = cat(grid, flow), sigmoid(certainty) (tiny.py:237-238) -> tiny_final_kernel
#include "tiny.h"

#include <algorithm>

namespace roma {

__device__ __forceinline__ float tiny_pix_coord(int i, int n) { return -1.f + (2.f * i + 1.f) / n; }  // linspace(-1+1/n, 1-1/n, n)

// NCHW f32 -> NHWC f32 (backbone outputs arrive in torch's layout)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C,
                                                           long HW) {
  const long total = (long)B * HW * C;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long r = idx / C;
    const long p = r % HW;
    const long b = r / HW;
    out[idx] = in[(b * C + c) * HW + p];
  }
}

int nchw_to_nhwc_launch(const float* in, float* out, int B, int C, int H, int W, hipStream_t s) {
  ROMA_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad arguments");
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 65536)), dim3(256), 0, s, in, out, B, C,
                     (long)H * W);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// pos_embed, inference path with the low-resolution softmax (tiny.py:123-137): for every query i = (h0, w0) of image A
//   best = argmax_j cv[b, j, i]                                    (j over the H1 x W1 positions of image B)
//   P    = softmax over { cv[b, (4y, 4x), i] } U { float(best) }    (the reference concatenates the INDEX tensor, tiny.py:134)
//   out  = sum_k P_k * grid_lr[k] + P_last * grid[best]
// cv is [B, H1*W1, N0] (N0 = H0*W0 queries, contiguous): one thread per query walks its column; loads are coalesced over
// the queries of a wave.  H1, W1 are multiples of 4 (the images are resized to multiples of 32, the features are stride 8).
__global__ __launch_bounds__(256) void tiny_pos_embed_kernel(const float* __restrict__ cv, float* __restrict__ out, int B, int H1,
                                                             int W1, long N0) {
  const long total = (long)B * N0;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long b = idx / N0, i = idx - b * N0;
  const int n1 = H1 * W1;
  const float* col = cv + b * (long)n1 * N0 + i;
  float best = col[0];
  int bi = 0;
  for (int j = 1; j < n1; ++j) {
    const float v = col[(long)j * N0];
    if (v > best) {  // first maximum, like torch.argmax
      best = v;
      bi = j;
    }
  }
  const int hl = H1 / 4, wl = W1 / 4;
  const float extra = (float)bi;
  float m = extra;
  for (int y = 0; y < hl; ++y)
    for (int x = 0; x < wl; ++x) m = fmaxf(m, col[(long)(4 * y * W1 + 4 * x) * N0]);
  float s = 0.f, px = 0.f, py = 0.f;
  for (int y = 0; y < hl; ++y) {
    const float gy = -1.f + (4.f * (2.f * y + 1.f)) / H1;  // linspace(-1 + 4/H1, 1 - 4/H1, H1/4)
    for (int x = 0; x < wl; ++x) {
      const float e = expf(col[(long)(4 * y * W1 + 4 * x) * N0] - m);
      s += e;
      px += e * (-1.f + (4.f * (2.f * x + 1.f)) / W1);
      py += e * gy;
    }
  }
  const float el = expf(extra - m);
  s += el;
  px += el * tiny_pix_coord(bi % W1, W1);
  py += el * tiny_pix_coord(bi / W1, H1);
  out[idx * 2 + 0] = px / s;
  out[idx * 2 + 1] = py / s;
}

__global__ void tiny_pos_embed_exact_kernel(const float* __restrict__ cv, float* __restrict__ out, int B, int H1, int W1, long N0);

int tiny_pos_embed_launch(const float* cv, float* out, int B, int H1, int W1, int H0, int W0, int exact_softmax, hipStream_t s) {
  ROMA_REQUIRE(cv && out && B > 0 && H1 > 0 && W1 > 0 && H0 > 0 && W0 > 0, "tiny_pos_embed: bad arguments");
  const long total = (long)B * H0 * W0;
  if (exact_softmax) {
    hipLaunchKernelGGL(tiny_pos_embed_exact_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cv, out, B, H1, W1,
                       (long)H0 * W0);
    ROMA_LAUNCH_CHECK();
    return 0;
  }
  ROMA_REQUIRE(H1 % 4 == 0 && W1 % 4 == 0, "tiny_pos_embed: the coarse feature map must be a multiple of 4 in both dimensions");
  hipLaunchKernelGGL(tiny_pos_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cv, out, B, H1, W1, (long)H0 * W0);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// d[b, y, x, :] = [ f0[b, y, x, 0:C] | bilinear_zeropad(f1[b], warp[b, y, x]) (C) | warp[b, y, x, 0:2] | 0 ... ]   (Cp channels)
// f0 [B, H, W, C], f1 [B, H1, W1, C] channels-last f32; warp [B, H, W, wc] (first two channels = normalised x, y).
__global__ __launch_bounds__(256) void tiny_matcher_input_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                                 const float* __restrict__ warp, int wc, float* __restrict__ d,
                                                                 int B, int H, int W, int H1, int W1, int C, int Cp) {
  const long total = (long)B * H * W * Cp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % Cp);
    const long pix = idx / Cp;
    float v = 0.f;
    if (c < C) {
      v = f0[pix * C + c];
    } else if (c < 2 * C) {
      const long b = pix / ((long)H * W);
      const float gx = warp[pix * wc + 0], gy = warp[pix * wc + 1];
      float ix = ((gx + 1.f) * W1 - 1.f) * 0.5f, iy = ((gy + 1.f) * H1 - 1.f) * 0.5f;
      ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
      iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
      const float fx0 = floorf(ix), fy0 = floorf(iy);
      const int x0 = (int)fx0, y0 = (int)fy0;
      const float tx = ix - fx0, ty = iy - fy0;
      const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
      const float* fb = f1 + b * (long)H1 * W1 * C + (c - C);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
        if (yy >= 0 && yy < H1 && xx >= 0 && xx < W1) v += wgt[t] * fb[((long)yy * W1 + xx) * C];
      }
    } else if (c < 2 * C + 2) {
      v = warp[pix * wc + (c - 2 * C)];
    }
    d[idx] = v;
  }
}

int tiny_matcher_input_launch(const float* f0, const float* f1, const float* warp, int warp_channels, float* d, int B, int H,
                              int W, int H1, int W1, int C, int Cp, hipStream_t s) {
  ROMA_REQUIRE(f0 && f1 && warp && d && B > 0 && H > 0 && W > 0 && H1 > 0 && W1 > 0, "tiny_matcher_input: bad arguments");
  ROMA_REQUIRE(warp_channels >= 2 && Cp >= 2 * C + 2, "tiny_matcher_input: Cp must hold 2 C + 2 channels");
  const long total = (long)B * H * W * Cp;
  hipLaunchKernelGGL(tiny_matcher_input_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 20)), dim3(256), 0, s, f0,
                     f1, warp, warp_channels, d, B, H, W, H1, W1, C, Cp);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// out[p, 0:3] = base[p, 0:nb] (missing channels = 0) + delta[p, 0:3] * (sx, sy, 1)      (tiny.py:289, 292, 300)
__global__ __launch_bounds__(256) void tiny_update_kernel(const float* __restrict__ base, int nb, const float* __restrict__ delta,
                                                          long ldd, float sx, float sy, float* __restrict__ out, long npix) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const float b0 = base[p * nb + 0], b1 = base[p * nb + 1], b2 = nb > 2 ? base[p * nb + 2] : 0.f;
  out[p * 3 + 0] = b0 + delta[p * ldd + 0] * sx;
  out[p * 3 + 1] = b1 + delta[p * ldd + 1] * sy;
  out[p * 3 + 2] = b2 + delta[p * ldd + 2];
}

int tiny_update_launch(const float* base, int base_channels, const float* delta, long ldd, float sx, float sy, float* out,
                       long npix, hipStream_t s) {
  ROMA_REQUIRE(base && delta && out && npix > 0 && (base_channels == 2 || base_channels == 3) && ldd >= 3,
               "tiny_update: bad arguments");
  hipLaunchKernelGGL(tiny_update_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, base, base_channels, delta, ldd, sx,
                     sy, out, npix);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// pos_embed with the exact softmax (exact_softmax=True / training branch, tiny.py:139-141):
//   P = softmax_j cv[b, j, i] over ALL H1 x W1 positions,  out = sum_j P_j * grid[j]
__global__ __launch_bounds__(256) void tiny_pos_embed_exact_kernel(const float* __restrict__ cv, float* __restrict__ out, int B,
                                                                   int H1, int W1, long N0) {
  const long total = (long)B * N0;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long b = idx / N0, i = idx - b * N0;
  const int n1 = H1 * W1;
  const float* col = cv + b * (long)n1 * N0 + i;
  float m = col[0];
  for (int j = 1; j < n1; ++j) m = fmaxf(m, col[(long)j * N0]);
  float s = 0.f, px = 0.f, py = 0.f;
  for (int y = 0; y < H1; ++y) {
    const float gy = tiny_pix_coord(y, H1);
    for (int x = 0; x < W1; ++x) {
      const float e = expf(col[(long)(y * W1 + x) * N0] - m);
      s += e;
      px += e * tiny_pix_coord(x, W1);
      py += e * gy;
    }
  }
  out[idx * 2 + 0] = px / s;
  out[idx * 2 + 1] = py / s;
}

// ================================================================================================ XFeat-style backbone
// TinyRoMa.forward_single (tiny.py:81-99) runs the caller's XFeat network: InstanceNorm2d(1) on the channel mean, then
// stacks of Conv2d (1x1 / 3x3, stride 1 / 2, with or without bias) + BatchNorm2d (eval) + ReLU, one AvgPool2d(4, 4), two
// bilinear resizes and two additions.  The Python side (roma_amd/tiny.py) walks the module once, folds every BatchNorm into
// its convolution and replays the layer list through the operators below - channels-last f32, no torch arithmetic.
// The whole backbone is < 1 GFLOP on images of a few hundred pixels (Cin <= 128), so these are plain direct kernels:
// coalesced, vectorised over 4 input / 4 output channels, nothing staged.

// gray = mean over the C input channels, then InstanceNorm2d(1) (no affine, biased variance, eps): one workgroup per image
__global__ __launch_bounds__(1024) void gray_instnorm_kernel(const float* __restrict__ in, float* __restrict__ out, int C, long HW,
                                                             float eps) {
  __shared__ float red[1024];
  __shared__ float stat;
  const float* ib = in + (long)blockIdx.x * HW * C;
  float* ob = out + (long)blockIdx.x * HW;
  const float invc = 1.f / (float)C;
  auto gray = [&](long p) {
    float g = 0.f;
    for (int c = 0; c < C; ++c) g += ib[p * C + c];
    return g * invc;
  };
  auto block_sum = [&](float v) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) stat = red[0];
    __syncthreads();
    return stat;
  };
  float acc = 0.f;
  for (long p = threadIdx.x; p < HW; p += 1024) acc += gray(p);
  const float mean = block_sum(acc) / (float)HW;
  acc = 0.f;
  for (long p = threadIdx.x; p < HW; p += 1024) {
    const float d = gray(p) - mean;
    acc += d * d;
  }
  const float inv = rsqrtf(block_sum(acc) / (float)HW + eps);
  for (long p = threadIdx.x; p < HW; p += 1024) ob[p] = (gray(p) - mean) * inv;
}

int gray_instnorm_launch(const float* in, float* out, int B, int H, int W, int C, float eps, hipStream_t s) {
  ROMA_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0, "gray_instnorm: bad arguments");
  hipLaunchKernelGGL(gray_instnorm_kernel, dim3((unsigned)B), dim3(1024), 0, s, in, out, C, (long)H * W, eps);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// out[b, yo, xo, co] = act(bias[co] + sum_{ky, kx, ci} in[b, yo S - P + ky, xo S - P + kx, ci] * w[(ky K + kx) Cin + ci][co]) + res
// One thread = one output pixel x 4 consecutive output channels; consecutive threads = consecutive channel groups of a pixel
// (weight reads coalesced, input reads broadcast).  Cout % 4 == 0.
template <bool VEC>
__global__ __launch_bounds__(256) void conv2d_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ res,
                                                          float* __restrict__ out, int B, int H, int W, int Cin, int Cout, int K,
                                                          int S, int P, int Ho, int Wo, int relu) {
  const int ng = Cout / 4;
  const long total = (long)B * Ho * Wo * ng;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int g = (int)(idx % ng);
    long r = idx / ng;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < K; ++ky) {
      const int yi = yo * S - P + ky;
      if (yi < 0 || yi >= H) continue;
      for (int kx = 0; kx < K; ++kx) {
        const int xi = xo * S - P + kx;
        if (xi < 0 || xi >= W) continue;
        const float* ip = in + (((long)b * H + yi) * W + xi) * Cin;
        const float* wp = w + (long)(ky * K + kx) * Cin * Cout + 4 * g;
        if (VEC) {
          for (int ci = 0; ci < Cin; ci += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ip + ci);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += a[j] * *reinterpret_cast<const f32x4*>(wp + (long)(ci + j) * Cout);
          }
        } else {
          for (int ci = 0; ci < Cin; ++ci) acc += ip[ci] * *reinterpret_cast<const f32x4*>(wp + (long)ci * Cout);
        }
      }
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
    }
    const long o = (((long)b * Ho + yo) * Wo + xo) * Cout + 4 * g;
    if (res) acc += *reinterpret_cast<const f32x4*>(res + o);
    *reinterpret_cast<f32x4*>(out + o) = acc;
  }
}

int conv2d_nhwc_launch(const float* in, const float* w, const float* bias, const float* res, float* out, int B, int H, int W,
                       int Cin, int Cout, int K, int stride, int pad, int relu, hipStream_t s) {
  ROMA_REQUIRE(in && w && out && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv2d_nhwc: bad arguments");
  ROMA_REQUIRE((K == 1 || K == 3) && (stride == 1 || stride == 2) && pad >= 0 && pad <= 1, "conv2d_nhwc: 1x1 / 3x3, stride 1 / 2, padding 0 / 1");
  ROMA_REQUIRE(Cout % 4 == 0, "conv2d_nhwc: Cout must be a multiple of 4");
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  ROMA_REQUIRE(Ho > 0 && Wo > 0, "conv2d_nhwc: empty output");
  const long total = (long)B * Ho * Wo * (Cout / 4);
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 1 << 20));
  if (Cin % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0)
    hipLaunchKernelGGL(conv2d_nhwc_kernel<true>, grid, dim3(256), 0, s, in, w, bias, res, out, B, H, W, Cin, Cout, K, stride, pad, Ho, Wo, relu);
  else
    hipLaunchKernelGGL(conv2d_nhwc_kernel<false>, grid, dim3(256), 0, s, in, w, bias, res, out, B, H, W, Cin, Cout, K, stride, pad, Ho, Wo, relu);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// AvgPool2d(k, stride k), floor output size, channels-last
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                           int C, int k, int Ho, int Wo) {
  const long total = (long)B * Ho * Wo * C;
  const float inv = 1.f / (float)(k * k);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float a = 0.f;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) a += in[(((long)b * H + yo * k + dy) * W + xo * k + dx) * C + c];
    out[idx] = a * inv;
  }
}

int avgpool_nhwc_launch(const float* in, float* out, int B, int H, int W, int C, int k, hipStream_t s) {
  ROMA_REQUIRE(in && out && B > 0 && C > 0 && k > 0 && H >= k && W >= k, "avgpool_nhwc: bad arguments");
  const int Ho = H / k, Wo = W / k;
  const long total = (long)B * Ho * Wo * C;
  hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 20)), dim3(256), 0, s, in, out, B, H,
                     W, C, k, Ho, Wo);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// out = a + b (+ c)
__global__ __launch_bounds__(256) void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                   float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = a[i] + b[i];  // (x3 + x4) + x5, the reference's order of evaluation (tiny.py:96)
    if (c) v += c[i];
    out[i] = v;
  }
}

int add3_launch(const float* a, const float* b, const float* c, float* out, long n, hipStream_t s) {
  ROMA_REQUIRE(a && b && out && n > 0, "add3: bad arguments");
  hipLaunchKernelGGL(add3_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1 << 20)), dim3(256), 0, s, a, b, c, out, n);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// warp[b, y, x] = (grid_x, grid_y, flow_x, flow_y), certainty = sigmoid(m[..., 2])      (tiny.py:226-238)
__global__ __launch_bounds__(256) void tiny_final_kernel(const float* __restrict__ m, float* __restrict__ warp,
                                                         float* __restrict__ cert, int B, int H, int W) {
  const long total = (long)B * H * W;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  const int y = (int)((idx / W) % H);
  *reinterpret_cast<f32x4*>(warp + idx * 4) = f32x4{tiny_pix_coord(x, W), tiny_pix_coord(y, H), m[idx * 3 + 0], m[idx * 3 + 1]};
  cert[idx] = 1.f / (1.f + expf(-m[idx * 3 + 2]));
}

int tiny_final_launch(const float* matches, float* warp, float* cert, int B, int H, int W, hipStream_t s) {
  ROMA_REQUIRE(matches && warp && cert && B > 0 && H > 0 && W > 0, "tiny_final: bad arguments");
  const long total = (long)B * H * W;
  hipLaunchKernelGGL(tiny_final_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, matches, warp, cert, B, H, W);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
