// HBM-bound kernels of the RoMa match() path for gfx950 (see elementwise.h).
// Conventions: channels-last activations, 16-byte (or 8-byte for bf16 quads) vector accesses,
// one wave64 per row for row reductions, f32 math everywhere.
#include <algorithm>
#include "elementwise.h"
#include "chol_diag.h"
#include <stdio.h>

#include <stdlib.h>
#include "gemm.h"  // DT_*

namespace roma {

#define ROMA_DT_SWITCH(dt, T, ...)            \
  if ((dt) == DT_F32) { using T = float; __VA_ARGS__; } else { using T = bf16_t; __VA_ARGS__; }

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// torch.linspace(-1+1/n, 1-1/n, n)[i] in f32 (symmetric two-sided evaluation like ATen)
__device__ inline float pix_coord(int i, int n) {
  const float start = (float)(-1.0 + 1.0 / n), end = (float)(1.0 - 1.0 / n);
  if (n == 1) return start;
  const float step = (end - start) / (float)(n - 1);
  return (i < n / 2) ? start + step * (float)i : end - step * (float)(n - 1 - i);
}

// ------------------------------------------------------------------ LayerNorm
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void layernorm_kernel(const TIN* x, const float* w, const float* b, TOUT* out,
                                                        long M, int D, float eps) {
  // one wave per row; a lane owns 8 consecutive elements of every 512-element chunk (two 16-byte loads, one 16-byte
  // bf16 store): the previous 4-element mapping wrote 8-byte pieces and ran at 2.9 TB/s
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const TIN* xr = x + row * D;
  f32x4 v[4][2];
  const int nv = D / 512;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv) {
      if constexpr (sizeof(TIN) == 2) {  // bf16 residual stream: one 16-byte load, widened exactly
        const uint4 r = *reinterpret_cast<const uint4*>(xr + i * 512 + lane * 8);
        const unsigned rp[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[i][j >> 1][2 * (j & 1)] = h16_lo(rp[j]);
          v[i][j >> 1][2 * (j & 1) + 1] = h16_hi(rp[j]);
        }
      } else {
        v[i][0] = *reinterpret_cast<const f32x4*>(xr + i * 512 + lane * 8);
        v[i][1] = *reinterpret_cast<const f32x4*>(xr + i * 512 + lane * 8 + 4);
      }
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv) s += (v[i][0][0] + v[i][0][1] + v[i][0][2] + v[i][0][3]) + (v[i][1][0] + v[i][1][1] + v[i][1][2] + v[i][1][3]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = v[i][u][j] - mean;
          q += d * d;
        }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv) {
      const int c = i * 512 + lane * 8;
      f32x4 o[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c + 4 * u);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + c + 4 * u);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[u][j] = (v[i][u][j] - mean) * rstd * wv[j] + bv[j];
      }
      if constexpr (sizeof(TOUT) == 2) {
        uint4 pk;
        pk.x = pack_bf16x2(o[0][0], o[0][1]);
        pk.y = pack_bf16x2(o[0][2], o[0][3]);
        pk.z = pack_bf16x2(o[1][0], o[1][1]);
        pk.w = pack_bf16x2(o[1][2], o[1][3]);
        *reinterpret_cast<uint4*>(out + row * D + c) = pk;
      } else {
        ElemIO<TOUT>::st4(out + row * D + c, o[0]);
        ElemIO<TOUT>::st4(out + row * D + c + 4, o[1]);
      }
    }
}

int layernorm_launch_dt(const void* x, int dt_in, const float* w, const float* b, void* out, long M, int D, float eps,
                        int dt_out, hipStream_t s) {
  ROMA_REQUIRE(D % 512 == 0 && D <= 2048, "layernorm: D must be a multiple of 512 and <= 2048");
  ROMA_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               "layernorm: x / out not 16-byte aligned");
  dim3 grid((unsigned)((M + 3) / 4));
  if (dt_in == DT_BF16) {
    ROMA_REQUIRE(dt_out == DT_BF16, "layernorm: bf16 input implies bf16 output");
    hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, w, b, (bf16_t*)out, M, D, eps);
  } else {
    ROMA_DT_SWITCH(dt_out, T, hipLaunchKernelGGL((layernorm_kernel<float, T>), grid, dim3(256), 0, s, (const float*)x, w, b,
                                                 (T*)out, M, D, eps));
  }
  ROMA_LAUNCH_CHECK();
  return 0;
}

int layernorm_launch(const float* x, const float* w, const float* b, void* out, long M, int D, float eps,
                     int dt_out, hipStream_t s) {
  return layernorm_launch_dt(x, DT_F32, w, b, out, M, D, eps, dt_out, s);
}

// ------------------------------------------------------------------ conv3x3, Cin = 3 (first VGG layer)
template <typename TOUT>
__global__ __launch_bounds__(256) void conv3x3_c3_kernel(const float* img, const float* w, const float* bias,
                                                         TOUT* out, int B, int H, int W) {
  __shared__ __attribute__((aligned(16))) float ws[27 * 64];
  __shared__ float bs[64];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) ws[i] = w[i];
  if (threadIdx.x < 64) bs[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const long HW = (long)H * W;
  const long pix = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  const int cg = threadIdx.x & 3;  // 16 output channels each
  if (pix >= (long)B * HW) return;
  const int b = (int)(pix / HW);
  const int rem = (int)(pix - (long)b * HW);
  const int y = rem / W, x = rem - y * W;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = bs[cg * 16 + j];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = img[((long)(b * 3 + ci) * H + yy) * W + xx];
        const float* wr = &ws[(ci * 9 + ky * 3 + kx) * 64 + cg * 16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
      }
  TOUT* o = out + pix * 64 + cg * 16;
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[j4 * 4 + j], 0.f);
    ElemIO<TOUT>::st4(o + j4 * 4, v);
  }
}

int conv3x3_c3_launch(const float* img, const float* w, const float* bias, void* out, int B, int H, int W,
                      int dt_out, hipStream_t s) {
  const long total = (long)B * H * W;
  dim3 grid((unsigned)((total + 63) / 64));
  ROMA_DT_SWITCH(dt_out, T, hipLaunchKernelGGL(conv3x3_c3_kernel<T>, grid, dim3(256), 0, s, img, w, bias, (T*)out, B, H, W));
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ im2col for the first 3x3 layer (K = 27 -> 32)
template <typename TOUT>
__global__ __launch_bounds__(256) void im2col3x3_c3_kernel(const float* img, TOUT* out, int B, int H, int W) {
  const long HW = (long)H * W;
  const long total = (long)B * HW * 8;  // 8 quads of 4 taps per pixel
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int quad = (int)(idx & 7);
    const long pix = idx >> 3;
    const int b = (int)(pix / HW);
    const int rem = (int)(pix - (long)b * HW);
    const int y = rem / W, x = rem - y * W;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = quad * 4 + j;
      float val = 0.f;
      if (k < 27) {
        const int ci = k / 9, t = k - ci * 9, ky = t / 3, kx = t - ky * 3;
        const int yy = y + ky - 1, xx = x + kx - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) val = img[((long)(b * 3 + ci) * H + yy) * W + xx];
      }
      v[j] = val;
    }
    ElemIO<TOUT>::st4(out + pix * 32 + quad * 4, v);
  }
}

int im2col3x3_c3_launch(const float* img, void* out, int B, int H, int W, int dt_out, hipStream_t s) {
  const long total = (long)B * H * W * 8;
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 1 << 20));
  ROMA_DT_SWITCH(dt_out, T, hipLaunchKernelGGL(im2col3x3_c3_kernel<T>, grid, dim3(256), 0, s, img, (T*)out, B, H, W));
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ MaxPool 2x2
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* in, T* out, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const long total = (long)B * Ho * Wo * C4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C4) * 4;
    long r = idx / C4;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const T* p = in + (((long)b * H + 2 * yo) * W + 2 * xo) * C + c;
    const f32x4 a0 = ElemIO<T>::ld4(p), a1 = ElemIO<T>::ld4(p + C), a2 = ElemIO<T>::ld4(p + (long)W * C),
                a3 = ElemIO<T>::ld4(p + (long)W * C + C);
    f32x4 m;
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a0[j], a1[j]), fmaxf(a2[j], a3[j]));
    ElemIO<T>::st4(out + (((long)b * Ho + yo) * Wo + xo) * C + c, m);
  }
}

int maxpool2x2_launch(const void* in, void* out, int B, int H, int W, int C, int dt, hipStream_t s) {
  ROMA_REQUIRE(C % 4 == 0, "maxpool: C must be a multiple of 4");
  const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 65536));
  ROMA_DT_SWITCH(dt, T, hipLaunchKernelGGL(maxpool_kernel<T>, grid, dim3(256), 0, s, (const T*)in, (T*)out, B, H, W, C));
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ DINOv2 patchify (im2col, 14x14 / stride 14)
template <typename TOUT>
__global__ __launch_bounds__(256) void im2col_patch14_kernel(const float* img, TOUT* out, int B, int H, int W, int Kpad) {
  const int th = H / 14, tw = W / 14;
  const long total = (long)B * th * tw * Kpad;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int k = (int)(idx % Kpad);
    long r = idx / Kpad;
    const int tx = (int)(r % tw);
    r /= tw;
    const int ty = (int)(r % th);
    const int b = (int)(r / th);
    float v = 0.f;
    if (k < 588) {
      const int c = k / 196, rem = k - c * 196, ky = rem / 14, kx = rem - ky * 14;
      v = img[((long)(b * 3 + c) * H + ty * 14 + ky) * W + tx * 14 + kx];
    }
    ElemIO<TOUT>::st(out + idx, v);
  }
}

int im2col_patch14_launch(const float* img, void* out, int B, int H, int W, int Kpad, int dt_out, hipStream_t s) {
  ROMA_REQUIRE(H % 14 == 0 && W % 14 == 0 && Kpad >= 588, "im2col_patch14: bad geometry");
  const long total = (long)B * (H / 14) * (W / 14) * Kpad;
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 65536));
  ROMA_DT_SWITCH(dt_out, T, hipLaunchKernelGGL(im2col_patch14_kernel<T>, grid, dim3(256), 0, s, img, (T*)out, B, H, W, Kpad));
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ token assembly (cls + pos-embed)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* patch, const float* cls, const float* pos,
                                                              float* tokens, int B, int T, int D) {
  const int D4 = D / 4;
  const long total = (long)B * (T + 1) * D4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % D4) * 4;
    const long r = idx / D4;
    const int t = (int)(r % (T + 1));
    const int b = (int)(r / (T + 1));
    f32x4 v = (t == 0) ? *reinterpret_cast<const f32x4*>(cls + c)
                       : *reinterpret_cast<const f32x4*>(patch + ((long)b * T + (t - 1)) * D + c);
    v += *reinterpret_cast<const f32x4*>(pos + (long)t * D + c);
    *reinterpret_cast<f32x4*>(tokens + r * D + c) = v;
  }
}

int assemble_tokens_launch(const float* patch, const float* cls, const float* pos, float* tokens, int B, int T,
                           int D, hipStream_t s) {
  const long total = (long)B * (T + 1) * (D / 4);
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 65536));
  hipLaunchKernelGGL(assemble_tokens_kernel, grid, dim3(256), 0, s, patch, cls, pos, tokens, B, T, D);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ strided copy / convert
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void copy2d_kernel(const TI* in, long ldi, TO* out, long ldo, long rows, int cols) {
  const int c4n = cols / 4;
  const long total = rows * c4n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % c4n) * 4;
    const long r = idx / c4n;
    ElemIO<TO>::st4(out + r * ldo + c, ElemIO<TI>::ld4(in + r * ldi + c));
  }
}

int copy2d_launch(const void* in, long ldi, int dt_in, void* out, long ldo, int dt_out, long rows, int cols,
                  hipStream_t s) {
  ROMA_REQUIRE(cols % 4 == 0 && ldi % 4 == 0 && ldo % 4 == 0, "copy2d: cols / strides must be multiples of 4");
  const long total = rows * (cols / 4);
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 65536));
  ROMA_DT_SWITCH(dt_in, TI, ROMA_DT_SWITCH(dt_out, TO, hipLaunchKernelGGL((copy2d_kernel<TI, TO>), grid, dim3(256), 0, s,
                                                                          (const TI*)in, ldi, (TO*)out, ldo, rows, cols)));
  ROMA_LAUNCH_CHECK();
  return 0;
}

// bfloat16 bits -> this build's 16-bit format (identity in the bf16 build): the hand-over of a ROMA_MIXED handle, whose
// DINOv2 ran in the bfloat16 library.  autocast does the same cast when the binary16 proj head consumes the bf16 features.
__global__ __launch_bounds__(256) void convert_from_bf16_kernel(const uint2* __restrict__ in, uint2* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const uint2 u = in[i];
    uint2 r;
    r.x = pack_bf16x2(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u));
    r.y = pack_bf16x2(__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    out[i] = r;
  }
}

int convert_from_bf16_launch(const void* in, void* out, long n, hipStream_t s) {
  ROMA_REQUIRE(n % 4 == 0 && n > 0, "convert_from_bf16: n must be a positive multiple of 4");
  const long n4 = n / 4;
  dim3 grid((unsigned)std::min<long>((n4 + 255) / 256, 65536));
  hipLaunchKernelGGL(convert_from_bf16_kernel, grid, dim3(256), 0, s, (const uint2*)in, (uint2*)out, n4);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ row L2 norms
template <typename T>
__global__ __launch_bounds__(256) void rownorm_kernel(const T* in, long ld, float* norms, long M, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = ElemIO<T>::ld4(in + row * ld + c);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = wave_sum(s);
  if (lane == 0) norms[row] = sqrtf(s);
}

int rownorm_launch(const void* in, long ld, int dt, float* norms, long M, int C, hipStream_t s) {
  ROMA_REQUIRE(C % 4 == 0, "rownorm: C must be a multiple of 4");
  dim3 grid((unsigned)((M + 3) / 4));
  ROMA_DT_SWITCH(dt, T, hipLaunchKernelGGL(rownorm_kernel<T>, grid, dim3(256), 0, s, (const T*)in, ld, norms, M, C));
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ GP Fourier basis (transposed)
// One copy per support image (blockIdx.y; copies `stride` floats apart): the GP stores F^T right behind each image's K_yy, and
// 16 back-to-back hipMemcpyAsync of one master copy cost more launches than recomputing 0.8 M cosines per image (round 5).
__global__ __launch_bounds__(256) void gp_basis_kernel(const float* w, const float* b, float* Ft, int Dg, int h, int wd,
                                                       int npad, long stride) {
  const long total = (long)Dg * npad;
  float* out = Ft + (long)blockIdx.y * stride;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int j = (int)(idx % npad);
    const int d = (int)(idx / npad);
    float v = 0.f;
    if (j < h * wd) {
      const int y = j / wd, x = j - y * wd;
      const float cx = pix_coord(x, wd), cy = pix_coord(y, h);
      const float z = w[d * 2 + 0] * cx + w[d * 2 + 1] * cy + b[d];
      v = cosf((float)(8.0 * 3.14159265358979323846) * z);
    }
    out[idx] = v;
  }
}

int gp_basis_launch(const float* w, const float* b, float* Ft, int Dg, int h, int wdt, int npad, hipStream_t s, int copies,
                    long stride) {
  const long total = (long)Dg * npad;
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 65536), (unsigned)std::max(copies, 1));
  hipLaunchKernelGGL(gp_basis_kernel, grid, dim3(256), 0, s, w, b, Ft, Dg, h, wdt, npad, stride);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ batched square transpose
__global__ __launch_bounds__(256) void transpose_kernel(const float* in, float* out, int n, long ld, long stride_in, long stride_out) {
  __shared__ float tile[32][33];
  const float* ib = in + (long)blockIdx.z * stride_in;
  float* ob = out + (long)blockIdx.z * stride_out;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8)
    if (y0 + j < n && x0 + tx < n) tile[j][tx] = ib[(long)(y0 + j) * ld + x0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (x0 + j < n && y0 + tx < n) ob[(long)(x0 + j) * ld + y0 + tx] = tile[tx][j];
}

int transpose_launch(const float* in, float* out, int n, long ld, int batch, hipStream_t s, long stride_in, long stride_out) {
  dim3 grid((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32), (unsigned)batch);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, in, out, n, ld, stride_in > 0 ? stride_in : (long)n * ld,
                     stride_out > 0 ? stride_out : (long)n * ld);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ 64x64 Cholesky + triangular inverse
// ONE wave64 per diagonal block: lane i owns row i of the factor (and later column i of the inverse) in LDS.
// A single wave needs no barriers (its LDS operations complete in order), pivots / multipliers are LDS broadcast
// reads, rows are stride-65 so lane-parallel accesses are conflict free.  The previous 256-thread version spent
// 127 us per call on 192 __syncthreads(); the blocked Cholesky issues 25 such calls on the GP's critical path.
// (A fully unrolled register/readlane variant was correct but ~70 KB of straight-line code: instruction-fetch bound.)
// Round 4: TWO waves in a pipeline.  Row r of the inverse only needs columns <= r of the factor, and column j is final the
// moment the factorisation has finished its step j - so wave 1 runs the forward substitutions one row behind wave 0's column
// steps instead of after them (wave 0 publishes its progress in LDS with a release store behind a step's writes, wave 1 polls
// it with an acquire load in front of a row's reads), and all four
// waves share the 16 KB load and the 48 KB write-back.  53 -> ~33 us per call on the GP's launch-latency-bound chain (25 calls);
// arithmetic and its order unchanged: bit-identical results.
__global__ __launch_bounds__(256) void chol_diag_kernel(float* A, long ld, long strideA, float* Linv, float* LinvT,
                                                        int k, int nblk) {
  // the factorisation / inverse wave pair and the write-back live in chol_diag.h (shared with chol_col.hip)
  constexpr int S = CHOL_S;
  __shared__ __attribute__((aligned(16))) float L[64 * S];   // L[i][c]   (lane i owns row i)
  __shared__ __attribute__((aligned(16))) float LT[64 * S];  // LT[j][c] = L[c][j]   (broadcast source)
  __shared__ __attribute__((aligned(16))) float XT[64 * S];  // XT[c][t] = X[t][c]   (lane c owns row c)
  __shared__ int progress;  // last finished column step of the factorisation (workgroup-scope release / acquire)
  float* Ab = A + (long)blockIdx.x * strideA + ((long)k * 64) * ld + (long)k * 64;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = tid & 63;
  for (int idx = tid; idx < 64 * 16; idx += 256) {  // coalesced float4 loads
    const int row = idx >> 4, c4 = idx & 15;
    *reinterpret_cast<f32x4*>(&L[row * S + c4 * 4]) = *reinterpret_cast<const f32x4*>(Ab + (long)row * ld + c4 * 4);
  }
  if (tid == 0) progress = -1;
  __syncthreads();
  if (wave == 0) chol_diag_factor_wave(L, LT, &progress, i);
  else if (wave == 1) chol_diag_inverse_wave(L, XT, &progress, i);
  __syncthreads();
  chol_diag_writeback(L, XT, Ab, ld, Linv + ((long)blockIdx.x * nblk + k) * 4096, LinvT + ((long)blockIdx.x * nblk + k) * 4096,
                      tid, 256);
}

int chol_diag_launch(float* A, long ld, long strideA, float* Linv, float* LinvT, int k, int nblk, int batch,
                     hipStream_t s) {
  hipLaunchKernelGGL(chol_diag_kernel, dim3((unsigned)batch), dim3(256), 0, s, A, ld, strideA, Linv, LinvT, k, nblk);
  ROMA_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void pad_identity_kernel(float* A, long ld, long strideA, int n, int npad) {
  float* Ab = A + (long)blockIdx.y * strideA;
  const long total = (long)npad * npad;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int i = (int)(idx / npad), j = (int)(idx % npad);
    if (i >= n || j >= n) Ab[(long)i * ld + j] = (i == j) ? 1.f : 0.f;
  }
}

int pad_identity_launch(float* A, long ld, long strideA, int n, int npad, int batch, hipStream_t s) {
  if (npad == n) return 0;
  const long total = (long)npad * npad;
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 4096), (unsigned)batch);
  hipLaunchKernelGGL(pad_identity_kernel, grid, dim3(256), 0, s, A, ld, strideA, n, npad);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ cls_to_flow_refine
__global__ __launch_bounds__(256) void cls_to_flow_kernel(const float* logits, long ld, float* flow, float* cert, long M) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* lr = logits + row * ld;
  float mx = -INFINITY;
  int arg = 0;
  f32x4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(lr + (i * 64 + lane) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[i][j] > mx) {
        mx = v[i][j];
        arg = (i * 64 + lane) * 4 + j;
      }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float om = __shfl_xor(mx, off);
    const int oa = __shfl_xor(arg, off);
    if (om > mx || (om == mx && oa < arg)) {
      mx = om;
      arg = oa;
    }
  }
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) se += expf(v[i][j] - mx);
  se = wave_sum(se);
  if (lane == 0) {
    const int idx5[5] = {arg - 1, arg, arg + 1, arg - 64, arg + 64};
    float fx = 0.f, fy = 0.f, tot = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int c = min(max(idx5[t], 0), 4095);
      const float p = expf(lr[c] - mx) / se;
      fx += p * pix_coord(c & 63, 64);
      fy += p * pix_coord(c >> 6, 64);
      tot += p;
    }
    flow[row * 2 + 0] = fx / tot;
    flow[row * 2 + 1] = fy / tot;
    cert[row] = lr[4096];
  }
}

int cls_to_flow_launch(const float* logits, long ld, float* flow, float* cert, long M, hipStream_t s) {
  ROMA_REQUIRE(ld % 4 == 0 && ld >= 4097, "cls_to_flow: bad leading dimension");
  dim3 grid((unsigned)((M + 3) / 4));
  hipLaunchKernelGGL(cls_to_flow_kernel, grid, dim3(256), 0, s, logits, ld, flow, cert, M);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ refiner input: [x | grid_sample(y, flow) | disp_emb | . | 0]
template <typename T>
__global__ __launch_bounds__(256) void refiner_input_kernel(const RefinerInputArgs a) {
  const int lane = threadIdx.x & 63;
  const long HW = (long)a.H * a.W;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (long)a.B * HW) return;
  const int b = (int)(pix / HW);
  const long p = pix - (long)b * HW;
  const int y = (int)(p / a.W), x = (int)(p - (long)y * a.W);
  const T* feat = reinterpret_cast<const T*>(a.feat);
  const T* fq = feat + ((long)b * HW + p) * a.ldf;
  const int simg = (b + a.shift) % a.nimg;
  const T* fs = feat + (long)simg * HW * a.ldf;
  T* d = reinterpret_cast<T*>(a.d) + pix * a.ldd;
  const float wx = a.flow[pix * 2 + 0], wy = a.flow[pix * 2 + 1];
  float ix = ((wx + 1.f) * a.W - 1.f) * 0.5f, iy = ((wy + 1.f) * a.H - 1.f) * 0.5f;
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
  iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float tx = ix - fx0, ty = iy - fy0;
  const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
  bool ok[4];
  long off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
    ok[t] = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    off[t] = ((long)yy * a.W + xx) * a.ldf;
  }
  for (int c = lane * 4; c < a.C; c += 256) {
    ElemIO<T>::st4(d + c, ElemIO<T>::ld4(fq + c));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (ok[t]) {
        const f32x4 v = ElemIO<T>::ld4(fs + off[t] + c);
        acc += wgt[t] * v;
      }
    ElemIO<T>::st4(d + a.C + c, acc);
  }
  const float dx = a.disp_scale * (wx - pix_coord(x, a.W)), dy = a.disp_scale * (wy - pix_coord(y, a.H));
  for (int e = lane; e < a.E; e += 64) {
    const float v = a.emb_w[e * 2 + 0] * dx + a.emb_w[e * 2 + 1] * dy + a.emb_b[e];
    ElemIO<T>::st(d + 2 * a.C + e, v);
  }
  for (long c = 2 * a.C + a.E + a.Kcorr + lane; c < a.ldd; c += 64) ElemIO<T>::st(d + c, 0.f);
}

// small / odd channel counts (stride-1 refiner: C = 9): one thread per (pixel, padded channel slot)
template <typename T>
__global__ __launch_bounds__(256) void refiner_input_small_kernel(const RefinerInputArgs a) {
  const long HW = (long)a.H * a.W;
  const int slots = (int)a.ldd;
  const long total = (long)a.B * HW * slots;
  const T* feat = reinterpret_cast<const T*>(a.feat);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % slots);
    const long pix = idx / slots;
    const int b = (int)(pix / HW);
    const long p = pix - (long)b * HW;
    T* d = reinterpret_cast<T*>(a.d) + pix * a.ldd;
    if (c < a.C) {
      ElemIO<T>::st(d + c, ElemIO<T>::ld(feat + ((long)b * HW + p) * a.ldf + c));
    } else if (c < 2 * a.C) {
      const int cc = c - a.C;
      const int simg = (b + a.shift) % a.nimg;
      const T* fs = feat + (long)simg * HW * a.ldf;
      const float wx = a.flow[pix * 2 + 0], wy = a.flow[pix * 2 + 1];
      float ix = ((wx + 1.f) * a.W - 1.f) * 0.5f, iy = ((wy + 1.f) * a.H - 1.f) * 0.5f;
      ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
      iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
      const float fx0 = floorf(ix), fy0 = floorf(iy);
      const int x0 = (int)fx0, y0 = (int)fy0;
      const float tx = ix - fx0, ty = iy - fy0;
      const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
          acc += wgt[t] * ElemIO<T>::ld(fs + ((long)yy * a.W + xx) * a.ldf + cc);
      }
      ElemIO<T>::st(d + c, acc);
    } else if (c < 2 * a.C + a.E) {
      const int e = c - 2 * a.C;
      const int y = (int)(p / a.W), x = (int)(p - (long)y * a.W);
      const float dx = a.disp_scale * (a.flow[pix * 2 + 0] - pix_coord(x, a.W));
      const float dy = a.disp_scale * (a.flow[pix * 2 + 1] - pix_coord(y, a.H));
      ElemIO<T>::st(d + c, a.emb_w[e * 2 + 0] * dx + a.emb_w[e * 2 + 1] * dy + a.emb_b[e]);
    } else if (c >= 2 * a.C + a.E + a.Kcorr) {
      ElemIO<T>::st(d + c, 0.f);
    }
  }
}

// stride-1 refiner (C = 9, E = 6, 24 output channels): one thread per pixel, whole rows as 16-byte vectors
template <typename T> struct VecIO;
template <> struct VecIO<float> {
  static constexpr int CV = 4;
  typedef f32x4 Raw;  // 16 bytes as loaded; cvt() turns them into CV floats (the same values ld() delivers)
  __device__ static inline Raw ldraw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  __device__ static inline void cvt(const Raw& x, float* v) { v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3]; }
  __device__ static inline void ld(const float* p, float* v) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(p);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  }
  __device__ static inline void st(float* p, const float* v) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct VecIO<bf16_t> {
  static constexpr int CV = 8;
  typedef uint4 Raw;
  __device__ static inline Raw ldraw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static inline void cvt(const Raw& u, float* v) {
    v[0] = h16_lo(u.x); v[1] = h16_hi(u.x);
    v[2] = h16_lo(u.y); v[3] = h16_hi(u.y);
    v[4] = h16_lo(u.z); v[5] = h16_hi(u.z);
    v[6] = h16_lo(u.w); v[7] = h16_hi(u.w);
  }
  __device__ static inline void ld(const bf16_t* p, float* v) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = h16_lo(u.x); v[1] = h16_hi(u.x);
    v[2] = h16_lo(u.y); v[3] = h16_hi(u.y);
    v[4] = h16_lo(u.z); v[5] = h16_hi(u.z);
    v[6] = h16_lo(u.w); v[7] = h16_hi(u.w);
  }
  __device__ static inline void st(bf16_t* p, const float* v) {
    uint4 u;
    u.x = pack_bf16x2(v[0], v[1]);
    u.y = pack_bf16x2(v[2], v[3]);
    u.z = pack_bf16x2(v[4], v[5]);
    u.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
  }
};

template <typename T, int C, int E>
__global__ __launch_bounds__(256) void refiner_input_pix_kernel(const RefinerInputArgs a) {
  constexpr int LDF = 16, LDD = 24;
  const long HW = (long)a.H * a.W;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (long)a.B * HW) return;
  const int b = (int)(pix / HW);
  const long p = pix - (long)b * HW;
  const int y = (int)(p / a.W), x = (int)(p - (long)y * a.W);
  const T* feat = reinterpret_cast<const T*>(a.feat);
  const T* fq = feat + ((long)b * HW + p) * LDF;
  const int simg = (b + a.shift) % a.nimg;
  const T* fs = feat + (long)simg * HW * LDF;
  if constexpr (sizeof(T) == 2) {
    // bf16: 16-byte accesses - a pixel's 16 feature channels are two loads (were three 8-byte ones), its 24 output channels
    // three stores (were six): this kernel is bound by the number of memory instructions, not by bytes
    static_assert(C <= 16 && 2 * C + E <= LDD && LDD == 24 && LDF == 16, "layout");
    float q[16], xh[16];
    VecIO<T>::ld(fq, q);
    VecIO<T>::ld(fq + 8, q + 8);
#pragma unroll
    for (int j = 0; j < 16; ++j) xh[j] = 0.f;
    const float wx = a.flow[pix * 2 + 0], wy = a.flow[pix * 2 + 1];
    float ix = ((wx + 1.f) * a.W - 1.f) * 0.5f, iy = ((wy + 1.f) * a.H - 1.f) * 0.5f;
    ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
    iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float tx = ix - fx0, ty = iy - fy0;
    const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
      if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
        const T* r = fs + ((long)yy * a.W + xx) * LDF;
        float v[16];
        VecIO<T>::ld(r, v);
        VecIO<T>::ld(r + 8, v + 8);
#pragma unroll
        for (int j = 0; j < C; ++j) xh[j] += wgt[t] * v[j];
      }
    }
    const float dx = a.disp_scale * (wx - pix_coord(x, a.W)), dy = a.disp_scale * (wy - pix_coord(y, a.H));
    float o[LDD];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      o[c] = q[c];
      o[C + c] = xh[c];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) o[2 * C + e] = a.emb_w[e * 2 + 0] * dx + a.emb_w[e * 2 + 1] * dy + a.emb_b[e];
#pragma unroll
    for (int c = 2 * C + E; c < LDD; ++c) o[c] = 0.f;
    T* d = reinterpret_cast<T*>(a.d) + pix * LDD;
#pragma unroll
    for (int c8 = 0; c8 < LDD / 8; ++c8) VecIO<T>::st(d + c8 * 8, o + c8 * 8);
    return;
  }
  float q[12], xh[12];
#pragma unroll
  for (int c4 = 0; c4 < 3; ++c4) {
    const f32x4 v = ElemIO<T>::ld4(fq + c4 * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q[c4 * 4 + j] = v[j];
      xh[c4 * 4 + j] = 0.f;
    }
  }
  const float wx = a.flow[pix * 2 + 0], wy = a.flow[pix * 2 + 1];
  float ix = ((wx + 1.f) * a.W - 1.f) * 0.5f, iy = ((wy + 1.f) * a.H - 1.f) * 0.5f;
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
  iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float tx = ix - fx0, ty = iy - fy0;
  const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
  // (guarded taps on purpose: the branch-free form that helps the vector kernel below - clamped address + select - was
  //  measured SLOWER here, 0.94 vs 0.78 ms per step)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
    if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
      const T* r = fs + ((long)yy * a.W + xx) * LDF;
#pragma unroll
      for (int c4 = 0; c4 < 3; ++c4) {
        const f32x4 v = ElemIO<T>::ld4(r + c4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) xh[c4 * 4 + j] += wgt[t] * v[j];
      }
    }
  }
  const float dx = a.disp_scale * (wx - pix_coord(x, a.W)), dy = a.disp_scale * (wy - pix_coord(y, a.H));
  float o[LDD];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    o[c] = q[c];
    o[C + c] = xh[c];
  }
#pragma unroll
  for (int e = 0; e < E; ++e) o[2 * C + e] = a.emb_w[e * 2 + 0] * dx + a.emb_w[e * 2 + 1] * dy + a.emb_b[e];
#pragma unroll
  for (int c = 2 * C + E; c < LDD; ++c) o[c] = 0.f;
  T* d = reinterpret_cast<T*>(a.d) + pix * LDD;
#pragma unroll
  for (int c4 = 0; c4 < LDD / 4; ++c4) ElemIO<T>::st4(d + c4 * 4, f32x4{o[c4 * 4], o[c4 * 4 + 1], o[c4 * 4 + 2], o[c4 * 4 + 3]});
}

// ------------------------------------------------------------------ depthwise 5x5 + BN(folded) + ReLU
// One thread = one 16-byte channel vector (4 f32 / 8 bf16) x 4 consecutive x positions of one row; the 5x8 input
// window is walked row by row so each loaded vector feeds up to 4 outputs.  Workgroups are remapped so that each
// XCD (private L2) owns a contiguous band of image rows: the 5-row vertical reuse then hits that XCD's L2 instead of
// being replicated in all eight.
// refiner input, vectorised: LPP lanes (a power of two) share one pixel, 64/LPP pixels per wave, 16 bytes per lane
// per access.  The wave-per-pixel kernel above left 48 of 64 lanes idle at C = 64 (stride-2 refiner, 4.2 M pixels).
template <typename T>
__global__ __launch_bounds__(256) void refiner_input_vec_kernel(const RefinerInputArgs a, int lpp) {
  constexpr int CV = VecIO<T>::CV;
  const int lane = threadIdx.x & 63;
  const int sub = lane & (lpp - 1), pw = lane / lpp, ppw = 64 / lpp;
  const long HW = (long)a.H * a.W;
  const long pix = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ppw + pw;
  if (pix >= (long)a.B * HW) return;
  const int b = (int)(pix / HW);
  const long p = pix - (long)b * HW;
  const int y = (int)(p / a.W), x = (int)(p - (long)y * a.W);
  const T* feat = reinterpret_cast<const T*>(a.feat);
  const T* fq = feat + ((long)b * HW + p) * a.ldf;
  const int simg = (b + a.shift) % a.nimg;
  const T* fs = feat + (long)simg * HW * a.ldf;
  T* d = reinterpret_cast<T*>(a.d) + pix * a.ldd;
  const float wx = a.flow[pix * 2 + 0], wy = a.flow[pix * 2 + 1];
  float ix = ((wx + 1.f) * a.W - 1.f) * 0.5f, iy = ((wy + 1.f) * a.H - 1.f) * 0.5f;
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
  iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float tx = ix - fx0, ty = iy - fy0;
  const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
  bool ok[4];
  long off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
    ok[t] = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;  // zeros padding: out-of-image taps are dropped (select below)
    off[t] = ((long)min(max(yy, 0), a.H - 1) * a.W + min(max(xx, 0), a.W - 1)) * a.ldf;  // always a valid address
  }
  for (int c = sub * CV; c < a.C; c += lpp * CV) {
    *reinterpret_cast<uint4*>(d + c) = *reinterpret_cast<const uint4*>(fq + c);
    float r[CV];
#pragma unroll
    for (int j = 0; j < CV; ++j) r[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // branch-free: the four tap loads of a channel piece are in flight together
      float tv[CV];
      VecIO<T>::ld(fs + off[t] + c, tv);
#pragma unroll
      for (int j = 0; j < CV; ++j) r[j] = ok[t] ? r[j] + wgt[t] * tv[j] : r[j];
    }
    VecIO<T>::st(d + a.C + c, r);
  }
  const float dx = a.disp_scale * (wx - pix_coord(x, a.W)), dy = a.disp_scale * (wy - pix_coord(y, a.H));
  for (int e = sub; e < a.E; e += lpp) {
    const float v = a.emb_w[e * 2 + 0] * dx + a.emb_w[e * 2 + 1] * dy + a.emb_b[e];
    ElemIO<T>::st(d + 2 * a.C + e, v);
  }
  for (long c = 2 * a.C + a.E + a.Kcorr + sub; c < a.ldd; c += lpp) ElemIO<T>::st(d + c, 0.f);
}

int refiner_input_launch(const RefinerInputArgs& a, hipStream_t s) {
  const long npix = (long)a.B * a.H * a.W;
  // grid_sample warp + concat writer (matcher.py:132-148,166).  Algorithmic bytes: x and the sampled y
  // read once (2 s C), the flow (8), the written row of d without the correlation slice another kernel fills
  const double es = a.dt == DT_F32 ? 4.0 : 2.0;
  char pname[64];
  snprintf(pname, sizeof pname, "refiner_input_warp<C=%d,%s>", a.C, a.dt == DT_F32 ? "f32" : ROMA_H16_NAME);
  ProfScope ps(pname, (double)npix * (2.0 * a.C * es + 8.0 + (double)(a.ldd - a.Kcorr) * es), "byte", s);
  if (a.C == 9 && a.E == 6 && a.Kcorr == 0 && a.ldf == 16 && a.ldd == 24) {
    dim3 grid((unsigned)((npix + 255) / 256));
    ROMA_DT_SWITCH(a.dt, T, hipLaunchKernelGGL((refiner_input_pix_kernel<T, 9, 6>), grid, dim3(256), 0, s, a));
    ROMA_LAUNCH_CHECK();
    return 0;
  }
  if (a.C % 4 != 0 || a.C < 32) {
    const long total = npix * a.ldd;
    dim3 grid((unsigned)std::min<long>((total + 255) / 256, 1 << 20));
    ROMA_DT_SWITCH(a.dt, T, hipLaunchKernelGGL(refiner_input_small_kernel<T>, grid, dim3(256), 0, s, a));
    ROMA_LAUNCH_CHECK();
    return 0;
  }
  ROMA_REQUIRE(a.ldf % 4 == 0 && a.ldd % 4 == 0, "refiner_input: strides must be multiples of 4");
  {
    const int cv = a.dt == DT_F32 ? 4 : 8;
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.feat) | reinterpret_cast<uintptr_t>(a.d)) & 15) == 0;
    static const bool vec_off = getenv("ROMA_RI_VEC") && atoi(getenv("ROMA_RI_VEC")) == 0;  // A/B debugging only
    if (!vec_off && aligned && a.C % cv == 0 && a.ldf % cv == 0 && a.ldd % cv == 0) {
      int lpp = 1;
      while (lpp < 64 && lpp * cv < a.C) lpp *= 2;
      const long per_wg = 4 * (64 / lpp);
      dim3 vgrid((unsigned)((npix + per_wg - 1) / per_wg));
      ROMA_DT_SWITCH(a.dt, T, hipLaunchKernelGGL(refiner_input_vec_kernel<T>, vgrid, dim3(256), 0, s, a, lpp));
      ROMA_LAUNCH_CHECK();
      return 0;
    }
  }
  dim3 grid((unsigned)((npix + 3) / 4));
  ROMA_DT_SWITCH(a.dt, T, hipLaunchKernelGGL(refiner_input_kernel<T>, grid, dim3(256), 0, s, a));
  ROMA_LAUNCH_CHECK();
  return 0;
}

typedef __attribute__((ext_vector_type(2))) float f32x2;

// Rolling-window form.  The stencil is NOT HBM-bound on MI355X when written naively: 25 MAC per 2-byte element put
// it on the VALU / vector-L1 path (a first version re-loaded the 25 weight vectors and 6 input rows per 8 outputs
// and ran at 1.4 TB/s with the texture path saturated).  Here one thread owns 4 channels x 4 x-positions and walks a
// strip of rows top to bottom:
//   * its 25 x 4 weights stay in registers for the whole strip (one load per thread, not per output),
//   * each input row is loaded ONCE (8 vectors) and feeds the 5 output rows it touches (5 rolling accumulator
//     slots that are rotated after every row), i.e. 200 packed-f32 FMAs (v_pk_fma_f32) per 8 loads,
//   * the next input row is prefetched while the current one is multiplied.
// Workgroups are remapped so each XCD owns a contiguous band of strips (private L2 sees the 4-row halo once).
template <typename T> struct RawVec;
template <> struct RawVec<float> {
  typedef f32x4 raw;
  __device__ static inline raw ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  __device__ static inline raw zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ static inline raw mask(raw r, uint32_t m) {  // bitwise AND keeps the load unconditional (no branch)
    return f32x4{__uint_as_float(__float_as_uint(r[0]) & m), __uint_as_float(__float_as_uint(r[1]) & m),
                 __uint_as_float(__float_as_uint(r[2]) & m), __uint_as_float(__float_as_uint(r[3]) & m)};
  }
  __device__ static inline void cvt(raw r, f32x2& a, f32x2& b) { a = f32x2{r[0], r[1]}; b = f32x2{r[2], r[3]}; }
  __device__ static inline void st(float* p, f32x2 a, f32x2 b) { *reinterpret_cast<f32x4*>(p) = f32x4{a[0], a[1], b[0], b[1]}; }
};
template <> struct RawVec<bf16_t> {
  typedef uint2 raw;
  __device__ static inline raw ld(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  __device__ static inline raw zero() { return make_uint2(0, 0); }
  __device__ static inline raw mask(raw r, uint32_t m) { return make_uint2(r.x & m, r.y & m); }
  __device__ static inline void cvt(raw r, f32x2& a, f32x2& b) {
    a = f32x2{h16_lo(r.x), h16_hi(r.x)};
    b = f32x2{h16_lo(r.y), h16_hi(r.y)};
  }
  __device__ static inline void st(bf16_t* p, f32x2 a, f32x2 b) {
    uint2 u;
    u.x = pack_bf16x2(a[0], a[1]);
    u.y = pack_bf16x2(b[0], b[1]);
    *reinterpret_cast<uint2*>(p) = u;
  }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void dwconv5x5_kernel(const T* in, T* out, const float* w, const float* bias, int B,
                                                           int H, int W, int Cp, int SY, int GC, int nchunk, int nxg,
                                                           int nblocks) {
  typedef typename RawVec<T>::raw raw_t;
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [25][GC*4] weights + [GC*4] bias of this channel chunk
  const int per_xcd = (nblocks + 7) / 8;
  const long lb = (long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lb >= nblocks) return;
  const int chunk = (int)(lb % nchunk);
  long r = lb / nchunk;
  const int xg = (int)(r % nxg);
  r /= nxg;
  const int yt = (H + SY - 1) / SY;
  const int ys = (int)(r % yt) * SY;
  const int b = (int)(r / yt);
  const int XQ = 256 / GC;
  const int cg = threadIdx.x % GC, xq = threadIdx.x / GC;
  const int c0 = chunk * GC * 4;
  constexpr int cw = 256;  // fixed LDS row stride: weight reads become base + immediate offset
  // stage this chunk's 25 x (GC*4) weights + bias as float4s; all loads are issued before the first LDS write
  // (a dword-at-a-time loop serialised 26 dependent global loads = ~40 us per workgroup)
  {
    const int q4 = GC;                   // float4 columns actually used per row
    const int nvec = 26 * q4;            // <= 26 * 64
    f32x4 tmp[7];
#pragma unroll
    for (int it = 0; it < 7; ++it) {
      const int i = threadIdx.x + 256 * it;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < nvec) {
        const int t = i / q4, cq = i - t * q4;
        const int ch = c0 + cq * 4;
        if (ch < Cp) v = *reinterpret_cast<const f32x4*>(t < 25 ? w + (long)t * Cp + ch : bias + ch);
      }
      tmp[it] = v;
    }
#pragma unroll
    for (int it = 0; it < 7; ++it) {
      const int i = threadIdx.x + 256 * it;
      if (i < nvec) {
        const int t = i / q4, cq = i - t * q4;
        *reinterpret_cast<f32x4*>(&wsm[t * cw + cq * 4]) = tmp[it];
      }
    }
  }
  __syncthreads();
  const int c = c0 + cg * 4;
  const int xb = (xg * XQ + xq) * 4;
  if (xq >= XQ || c >= Cp || xb >= W) return;
  const int sy = min(SY, H - ys);

  const f32x4 bx = *reinterpret_cast<const f32x4*>(&wsm[25 * cw + cg * 4]);
  const f32x2 bias0 = f32x2{bx[0], bx[1]}, bias1 = f32x2{bx[2], bx[3]};
  f32x2 acc[5][4][2];
#pragma unroll
  for (int s5 = 0; s5 < 5; ++s5)
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      acc[s5][px][0] = bias0;
      acc[s5][px][1] = bias1;
    }
  bool colok[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) colok[j] = (xb - 2 + j >= 0) && (xb - 2 + j < W);
  long colofs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) colofs[j] = (long)min(max(xb - 2 + j, 0), W - 1) * Cp;
  const T* base = in + ((long)b * H * W) * Cp + c;
  T* obase = out + ((long)b * H * W) * Cp + c;

// branch-free row load: out-of-image taps read a clamped (valid) address and are zeroed by a select afterwards
#define ROMA_DW_LOAD_ROW(TT, DST)                                                                         \
  {                                                                                                       \
    const int yy_ = ys - 2 + (TT);                                                                        \
    const bool rok_ = yy_ >= 0 && yy_ < H;                                                                \
    const T* rowp_ = base + (long)min(max(yy_, 0), H - 1) * W * Cp;                                       \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                       \
      DST[j] = RawVec<T>::mask(RawVec<T>::ld(rowp_ + colofs[j]), (rok_ && colok[j]) ? 0xffffffffu : 0u);  \
    }                                                                                                     \
  }

  // acc[k] holds output row o = t - 4 + k while input row t is processed (tap row ky = 4 - k)
  raw_t cur[8], nxt[8];
  ROMA_DW_LOAD_ROW(0, cur);
#pragma nounroll
  for (int t = 0; t < sy + 4; ++t) {
    ROMA_DW_LOAD_ROW(t + 1, nxt);
    f32x2 v[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) RawVec<T>::cvt(cur[j], v[j][0], v[j][1]);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int ky = 4 - k;
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        const f32x4 wx = *reinterpret_cast<const f32x4*>(&wsm[(ky * 5 + kx) * cw + cg * 4]);
        const f32x2 w0 = f32x2{wx[0], wx[1]}, w1 = f32x2{wx[2], wx[3]};
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          acc[k][px][0] = v[px + kx][0] * w0 + acc[k][px][0];
          acc[k][px][1] = v[px + kx][1] * w1 + acc[k][px][1];
        }
      }
    }
    const int o = t - 4;
    if (o >= 0) {
      T* orow = obase + ((long)(ys + o) * W + xb) * Cp;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        if (xb + px < W) {
          f32x2 a0 = acc[0][px][0], a1 = acc[0][px][1];
          a0 = f32x2{fmaxf(a0[0], 0.f), fmaxf(a0[1], 0.f)};
          a1 = f32x2{fmaxf(a1[0], 0.f), fmaxf(a1[1], 0.f)};
          RawVec<T>::st(orow + (long)px * Cp, a0, a1);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        acc[k][px][0] = acc[k + 1][px][0];
        acc[k][px][1] = acc[k + 1][px][1];
      }
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      acc[4][px][0] = bias0;
      acc[4][px][1] = bias1;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
  }
}

#undef ROMA_DW_LOAD_ROW

int dwconv5x5_launch(const void* in, void* out, const float* w, const float* bias, int B, int H, int W, int Cp,
                     int dt, hipStream_t s) {
  ROMA_REQUIRE(Cp % 4 == 0, "dwconv5x5: padded channel count must be a multiple of 4");
  ROMA_REQUIRE(in != out, "dwconv5x5: in and out must not alias");
  {  // wide 16-bit problems: the wave-private LDS-DMA ring form (dwconv_ring.hip), bit-identical
    const int rc = dwconv5x5_ring_try_launch(in, out, w, bias, B, H, W, Cp, dt, s);
    if (rc <= 0) return rc;
  }
  const int CG = Cp / 4;
  const int nchunk = (CG + 63) / 64;
  const int GC = (CG + nchunk - 1) / nchunk;  // channel groups (of 4) per workgroup, <= 64
  const int XQ = 256 / GC;                    // x tiles (of 4 pixels) per workgroup
  const int nxg = ((W + 3) / 4 + XQ - 1) / XQ;
  // strip height: SY + 4 input rows are read per strip, so 36-row strips cost 11 % extra rows against 25 % for
  // 16-row strips - as long as there are still enough workgroups to fill 256 CUs x 2
  int SY = 36;
  if ((long)B * ((H + 35) / 36) * nxg * nchunk < 1024) SY = 16;
  const long nb = (long)B * ((H + SY - 1) / SY) * nxg * nchunk;
  const int nblocks = (int)nb;
  dim3 grid((unsigned)(((nblocks + 7) / 8) * 8));
  const size_t lds = (size_t)26 * 256 * sizeof(float);  // fixed 256-float rows (see kernel)
  ProfScope ps(dt == DT_F32 ? "dwconv5x5_kernel<f32>" : "dwconv5x5_kernel<" ROMA_H16_NAME ">",
               2.0 * (double)B * H * W * Cp * (dt == DT_F32 ? 4.0 : 2.0), "byte", s);
  ROMA_DT_SWITCH(dt, T, hipLaunchKernelGGL(dwconv5x5_kernel<T>, grid, dim3(256), lds, s, (const T*)in, (T*)out, w, bias, B, H, W, Cp, SY, GC, nchunk, nxg, nblocks));
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ out_conv + flow / certainty update
template <typename T, int G>
__global__ __launch_bounds__(256) void refiner_out_kernel(const T* d, long ldd, const float* w, const float* bb, float* flow,
                                                          float* cert, long M, int Cp, float sx, float sy) {
  const int lane = threadIdx.x & 63;
  const int sub = lane % G, rw = lane / G;
  constexpr int RPW = 64 / G;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + rw;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  if (row < M) {
    for (int c = sub * 4; c < Cp; c += G * 4) {
      const f32x4 v = ElemIO<T>::ld4(d + row * ldd + c);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + c);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + Cp + c);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(w + 2 * Cp + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = fmaf(v[j], w0[j], a0);
        a1 = fmaf(v[j], w1[j], a1);
        a2 = fmaf(v[j], w2[j], a2);
      }
    }
  }
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) {
    a0 += __shfl_xor(a0, off);
    a1 += __shfl_xor(a1, off);
    a2 += __shfl_xor(a2, off);
  }
  if (row < M && sub == 0) {
    flow[row * 2 + 0] += sx * (a0 + bb[0]);
    flow[row * 2 + 1] += sy * (a1 + bb[1]);
    cert[row] += a2 + bb[2];
  }
}

// Vectorised form: LPR lanes (a power of two) share a row, 16 bytes per lane per access, and a wave walks ROWS_IT
// groups of 64/LPR rows with its out_conv weights held in registers.  (The kernel above re-read 48 bytes of f32
// weights from L1/L2 for every 8 bytes of activations and left 28 of 64 lanes idle at Cp = 144: 4x off the HBM bound.)
template <typename T, int NK>
__global__ __launch_bounds__(256) void refiner_out_vec_kernel(const T* d, long ldd, const float* w, const float* bb,
                                                              float* flow, float* cert, long M, int Cp, float sx, float sy,
                                                              int lpr, int rows_it) {
  constexpr int CV = VecIO<T>::CV;
  const int lane = threadIdx.x & 63;
  const int sub = lane & (lpr - 1), rw = lane / lpr, rpw = 64 / lpr;
  // the lane's 3 x NK x CV out_conv weights, fetched as 16-byte pieces with a clamped index and selected afterwards (Cp is a
  // multiple of CV, refiner_out_launch checks w's alignment).  As 120 guarded 4-byte loads - a branch around each - this
  // prologue took a third of a workgroup's life at Cp = 576 (round 6, late: tools/bench_refiner_out.py).
  float wr[NK][3][CV];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int c = (sub + k * lpr) * CV;
    const bool okc = c < Cp;
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int j4 = 0; j4 < CV; j4 += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(w + (long)o * Cp + (okc ? c : 0) + j4);
#pragma unroll
        for (int u = 0; u < 4; ++u) wr[k][o][j4 + u] = okc ? t[u] : 0.f;
      }
  }
  const float b0 = bb[0], b1 = bb[1], b2 = bb[2];
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw * rows_it + rw;
  // Round 6 (late): the row groups are software-pipelined - the 16-byte pieces (and flow / certainty) of group it + 1 are
  // requested BEFORE the ~240 VALU instructions and the shuffle reduction of group it, into a second register set (the loop
  // runs in pairs, so the two sets swap roles without moves).  The `#pragma unroll 4` this replaces never happened (rows_it
  // is a run-time value): a wave had 5 x 16 bytes per lane in flight, waited, computed with nothing in flight - 1.9 TB/s
  // at Cp = 576 / 1152 (profiles/r06_final_bench_mixed_kernel_stats.csv of commit 49f591d: 0.98 ms per step for 1.83 GB).
  // Same loads, same arithmetic in the same order: bit-identical.
  typedef typename VecIO<T>::Raw Raw;
  struct Group {
    Raw v[NK];
    float f0, f1, c0;
  };
  auto fetch = [&](Group& g, long row) __attribute__((always_inline)) {
    // branch-free loads (clamped row and chunk, select afterwards): inside exec-masked blocks hipcc waits for every load on
    // its own, and this kernel is nothing but loads (1.5 TB/s at Cp = 144 with the guarded form)
    const long rowc = row < M ? row : M - 1;
    // the running flow / certainty are fetched with the activations, not after the reduction (a dependent
    // load -> add -> store tail per row group otherwise)
    g.f0 = flow[rowc * 2 + 0];
    g.f1 = flow[rowc * 2 + 1];
    g.c0 = cert[rowc];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int c = (sub + k * lpr) * CV;
      g.v[k] = VecIO<T>::ldraw(d + rowc * ldd + (c < Cp ? c : 0));
    }
  };
  auto finish = [&](const Group& g, long row) __attribute__((always_inline)) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      float v[CV];
      VecIO<T>::cvt(g.v[k], v);
      const bool ok = (sub + k * lpr) * CV < Cp;
#pragma unroll
      for (int j = 0; j < CV; ++j) {
        const float x = ok ? v[j] : 0.f;
        a0 = fmaf(x, wr[k][0][j], a0);
        a1 = fmaf(x, wr[k][1][j], a1);
        a2 = fmaf(x, wr[k][2][j], a2);
      }
    }
    for (int off = lpr >> 1; off >= 1; off >>= 1) {
      a0 += __shfl_xor(a0, off);
      a1 += __shfl_xor(a1, off);
      a2 += __shfl_xor(a2, off);
    }
    if (row < M && sub == 0) {
      flow[row * 2 + 0] = g.f0 + sx * (a0 + b0);
      flow[row * 2 + 1] = g.f1 + sy * (a1 + b1);
      cert[row] = g.c0 + (a2 + b2);
    }
  };
  Group ga, gb;
  fetch(ga, row0);
  for (int it = 0; it < rows_it; it += 2) {  // rows_it is even (refiner_out_launch)
    const long row = row0 + (long)it * rpw;
    fetch(gb, row + rpw);
    finish(ga, row);
    if (it + 2 < rows_it) fetch(ga, row + 2 * rpw);  // uniform
    finish(gb, row + rpw);
  }
}

// Narrow rows (bf16, Cp = 24: the stride-1 refiner, 12 M pixels per call): ONE LANE PER ROW.  The shared-row kernel above
// gives such a row to two lanes (3 chunks: the second lane idles in the second piece, 2.8 TB/s); here a lane reads its
// row's NCH 16-byte chunks itself (the 48-byte lane stride leaves every fetched line fully used across the NCH loads,
// L1 serves the repeats), keeps all 3 x Cp weights in registers over ROWS rows, needs no shuffles, and the running
// flow / certainty are lane-contiguous 8- and 4-byte accesses.
template <int NCH, int ROWS>
__global__ __launch_bounds__(256) void refiner_out_row_kernel(const bf16_t* d, long ldd, const float* w, const float* bb,
                                                              float* flow, float* cert, long M, int Cp, float sx, float sy) {
  float wr[3][NCH * 8];
#pragma unroll
  for (int o = 0; o < 3; ++o)
#pragma unroll
    for (int j = 0; j < NCH * 8; ++j) wr[o][j] = j < Cp ? w[(long)o * Cp + j] : 0.f;
  const float b0 = bb[0], b1 = bb[1], b2 = bb[2];
  const long row0 = (long)blockIdx.x * 256 * ROWS + threadIdx.x;
#pragma unroll 2
  for (int it = 0; it < ROWS; ++it) {
    const long row = row0 + (long)it * 256;
    if (row >= M) break;
    const f32x2 f = *reinterpret_cast<const f32x2*>(flow + row * 2);
    const float c0 = cert[row];
    float v[NCH][8];
#pragma unroll
    for (int k = 0; k < NCH; ++k) VecIO<bf16_t>::ld(d + row * ldd + k * 8, v[k]);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0 = fmaf(v[k][j], wr[0][k * 8 + j], a0);
        a1 = fmaf(v[k][j], wr[1][k * 8 + j], a1);
        a2 = fmaf(v[k][j], wr[2][k * 8 + j], a2);
      }
    *reinterpret_cast<f32x2*>(flow + row * 2) = f32x2{f[0] + sx * (a0 + b0), f[1] + sy * (a1 + b1)};
    cert[row] = c0 + (a2 + b2);
  }
}

// flow / certainty += the per-pixel deltas the FINAL refiner blocks wrote (refiner_block.h): flow [M][2], cert [M], delta [M][4]
__global__ __launch_bounds__(256) void refiner_apply_delta_kernel(const f32x4* __restrict__ delta, float* __restrict__ flow,
                                                                  float* __restrict__ cert, long M, float sx, float sy) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long)gridDim.x * 256) {
    const f32x4 d = delta[i];
    f32x2 f = *reinterpret_cast<const f32x2*>(flow + 2 * i);
    f[0] += sx * d[0];
    f[1] += sy * d[1];
    *reinterpret_cast<f32x2*>(flow + 2 * i) = f;
    cert[i] += d[2];
  }
}

int refiner_apply_delta_launch(const float* delta, float* flow, float* cert, long M, float sx, float sy, hipStream_t s) {
  ROMA_REQUIRE((reinterpret_cast<uintptr_t>(delta) & 15) == 0 && (reinterpret_cast<uintptr_t>(flow) & 7) == 0, "refiner_apply_delta: alignment");
  const long nb = std::min<long>((M + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(refiner_apply_delta_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const f32x4*)delta, flow, cert, M, sx, sy);
  ROMA_LAUNCH_CHECK();
  return 0;
}

int refiner_out_launch(const void* d, long ldd, int dt, const float* w, const float* b, float* flow, float* cert,
                       long M, int Cp, float sx, float sy, hipStream_t s) {
  ROMA_REQUIRE(Cp % 4 == 0 && ldd % 4 == 0, "refiner_out: channel padding must be a multiple of 4");
  static const int row_env = getenv("ROMA_OUT_ROW") ? atoi(getenv("ROMA_OUT_ROW")) : 1;
  if (row_env && dt == DT_BF16 && Cp == 24 && ldd % 8 == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(flow) & 7) == 0) {
    constexpr int ROWS = 8;
    dim3 grid((unsigned)((M + 256l * ROWS - 1) / (256l * ROWS)));
    hipLaunchKernelGGL((refiner_out_row_kernel<3, ROWS>), grid, dim3(256), 0, s, (const bf16_t*)d, ldd, w, b, flow, cert, M, Cp, sx, sy);
    ROMA_LAUNCH_CHECK();
    return 0;
  }
  {
    const int cv = dt == DT_F32 ? 4 : 8;
    const int chunks = Cp / cv;
    if (Cp % cv == 0 && ldd % cv == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 && chunks >= 1) {
      int lpr = 1;
      while (lpr < 64 && lpr * 2 <= chunks) lpr *= 2;       // largest power of two <= chunks (<= 64)
      // bf16: chunk counts just above a power of two (18, 72, 144) leave 44 % of the lane-pieces of that split empty (the
      // second piece has 2 / 8 / 16 active lanes of 16 / 64 / 64): take the power of two that fills the pieces best with
      // at most 5 per lane (18 = 4 x 5 - 2, 72 = 16 x 5 - 8, 144 = 32 x 5 - 16: 90 %), which also shortens the shuffle
      // reduction.  The f32 mode keeps its split (bit-for-bit history of the parity runs).
      static const int lpr_env = getenv("ROMA_OUT_LPR") ? atoi(getenv("ROMA_OUT_LPR")) : 1;
      if (dt == DT_BF16 && lpr_env) {
        int best_l = lpr;
        double best_e = (double)chunks / ((double)lpr * ((chunks + lpr - 1) / lpr));
        for (int l = lpr / 2; l >= 2; l /= 2) {
          const int n = (chunks + l - 1) / l;
          if (n > 5) break;
          const double e = (double)chunks / ((double)l * n);
          if (e > best_e + 1e-9) {
            best_e = e;
            best_l = l;
          }
        }
        lpr = best_l;
      }
      const int nk = (chunks + lpr - 1) / lpr;              // 16-byte pieces per lane per row
      if (nk <= 6) {
        // 8 row groups per wave.  (With the weight prologue as guarded scalar loads 32 groups were 1.5 x faster than 8; with the
        // 16-byte prologue 8 and 16 are equal and 32 loses on the sub-batch sizes - profiles/r06_v39_*.  ROMA_OUT_ROWS_IT: A/B, even.)
        static const int rows_it_env = getenv("ROMA_OUT_ROWS_IT") ? atoi(getenv("ROMA_OUT_ROWS_IT")) : 0;
        const int rows_it = rows_it_env >= 2 ? rows_it_env & ~1 : 8;
        const long rows_per_block = 4l * (64 / lpr) * rows_it;
        dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block));
#define ROMA_ROV(NKV)                                                                                          \
  ROMA_DT_SWITCH(dt, T, hipLaunchKernelGGL((refiner_out_vec_kernel<T, NKV>), grid, dim3(256), 0, s, (const T*)d, ldd, w, b, \
                                           flow, cert, M, Cp, sx, sy, lpr, rows_it))
        if (nk <= 1) { ROMA_ROV(1); }
        else if (nk == 2) { ROMA_ROV(2); }
        else if (nk == 3) { ROMA_ROV(3); }
        else if (nk == 5) { ROMA_ROV(5); }
        else { ROMA_ROV(6); }
#undef ROMA_ROV
        ROMA_LAUNCH_CHECK();
        return 0;
      }
    }
  }
  const int chunks = Cp / 4;
#define ROMA_RO(G)                                                                                          \
  {                                                                                                         \
    const long rows_per_block = 4 * (64 / G);                                                               \
    dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block));                                       \
    ROMA_DT_SWITCH(dt, T, hipLaunchKernelGGL((refiner_out_kernel<T, G>), grid, dim3(256), 0, s, (const T*)d, ldd, w, b, \
                                             flow, cert, M, Cp, sx, sy));                                   \
  }
  if (chunks <= 8) ROMA_RO(8)
  else if (chunks <= 16) ROMA_RO(16)
  else if (chunks <= 32) ROMA_RO(32)
  else ROMA_RO(64)
#undef ROMA_RO
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ bilinear resize (align_corners=False)
__device__ inline void bilinear_src(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* in, float* out, int B, int Hin, int Win, int Hout,
                                                              int Wout, int nc) {
  const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
  const long total = (long)B * Hout * Wout;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = (int)(idx % Wout);
    long r = idx / Wout;
    const int y = (int)(r % Hout);
    const int b = (int)(r / Hout);
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_src(y, sh, Hin, y0, y1, ly);
    bilinear_src(x, sw, Win, x0, x1, lx);
    const float* ib = in + (long)b * Hin * Win * nc;
    for (int c = 0; c < nc; ++c) {
      const float v00 = ib[((long)y0 * Win + x0) * nc + c], v01 = ib[((long)y0 * Win + x1) * nc + c];
      const float v10 = ib[((long)y1 * Win + x0) * nc + c], v11 = ib[((long)y1 * Win + x1) * nc + c];
      out[idx * nc + c] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
  }
}

int resize_bilinear_launch(const float* in, float* out, int B, int Hin, int Win, int Hout, int Wout, int nc,
                           hipStream_t s) {
  const long total = (long)B * Hout * Wout;
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 65536));
  hipLaunchKernelGGL(resize_bilinear_kernel, grid, dim3(256), 0, s, in, out, B, Hin, Win, Hout, Wout, nc);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ final epilogue (matcher.py:839-850, 891-929)
__global__ __launch_bounds__(256) void final_epilogue_kernel(const FinalArgs a) {
  const int Wout = a.symmetric ? 2 * a.W : a.W;
  const long total = (long)a.B * a.H * Wout;
  const float sh = a.cert16 ? (float)a.h16 / (float)a.H : 0.f, sw = a.cert16 ? (float)a.w16 / (float)a.W : 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int xo = (int)(idx % Wout);
    long r = idx / Wout;
    const int y = (int)(r % a.H);
    const int bo = (int)(r / a.H);
    const int half = xo / a.W, x = xo - half * a.W;
    const int dp = bo + half * a.B;
    const long src = ((long)dp * a.H + y) * a.W + x;
    float fx = a.flow[src * 2 + 0], fy = a.flow[src * 2 + 1];
    float c = a.cert[src];
    if (a.cert16) {
      int y0, y1, x0, x1;
      float ly, lx;
      bilinear_src(y, sh, a.h16, y0, y1, ly);
      bilinear_src(x, sw, a.w16, x0, x1, lx);
      const float* cb = a.cert16 + (long)dp * a.h16 * a.w16;
      const float low = (1.f - ly) * ((1.f - lx) * cb[y0 * a.w16 + x0] + lx * cb[y0 * a.w16 + x1]) +
                        ly * ((1.f - lx) * cb[y1 * a.w16 + x0] + lx * cb[y1 * a.w16 + x1]);
      c -= (low < 0.f) ? 0.5f * low : 0.f;
    }
    c = 1.f / (1.f + expf(-c));
    if (fabsf(fx) > 1.f || fabsf(fy) > 1.f) c = 0.f;
    fx = fminf(fmaxf(fx, -1.f), 1.f);
    fy = fminf(fmaxf(fy, -1.f), 1.f);
    const float gx = pix_coord(x, a.W), gy = pix_coord(y, a.H);
    f32x4 o;
    if (half == 0) o = f32x4{gx, gy, fx, fy};
    else o = f32x4{fx, fy, gx, gy};
    *reinterpret_cast<f32x4*>(a.warp + idx * 4) = o;
    a.certainty[idx] = c;
  }
}

int final_epilogue_launch(const FinalArgs& a, hipStream_t s) {
  const long total = (long)a.B * a.H * (a.symmetric ? 2 * a.W : a.W);
  dim3 grid((unsigned)std::min<long>((total + 255) / 256, 65536));
  hipLaunchKernelGGL(final_epilogue_kernel, grid, dim3(256), 0, s, a);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- determinism trace (debug)
// Order-independent 64-bit checksum of a buffer: XOR over the words of mix(word, index).  Any single changed bit changes
// the sum; the result does not depend on how threads are scheduled.  Used by Model::trace (option "trace") to find the
// FIRST stage whose output differs between two runs of the same inputs.
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* p, long nwords, unsigned long long* out) {
  unsigned long long h = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (long)gridDim.x * 256) {
    unsigned long long x = ((unsigned long long)p[i] << 32) ^ (unsigned long long)(i * 0x9E3779B97F4A7C15ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    h ^= x;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)h, off), hi = __shfl_xor((unsigned)(h >> 32), off);
    h ^= ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0 && h) atomicXor(out, h);
}

// diagnostic variant: only columns [c0, c1) of a [rows][ld] matrix of 16-bit elements
__global__ __launch_bounds__(256) void checksum_cols_kernel(const unsigned short* p, long rows, int ld, int c0, int c1,
                                                            unsigned long long* out) {
  unsigned long long h = 0;
  const int nc = c1 - c0;
  const long n = rows * nc;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / nc;
    const int c = c0 + (int)(i - r * nc);
    unsigned long long x = ((unsigned long long)p[r * ld + c] << 32) ^ (unsigned long long)(i * 0x9E3779B97F4A7C15ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    h ^= x;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)h, off), hi = __shfl_xor((unsigned)(h >> 32), off);
    h ^= ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0 && h) atomicXor(out, h);
}

int checksum_cols_launch(const void* p, long rows, int ld, int c0, int c1, unsigned long long* out, hipStream_t s) {
  if (rows <= 0 || c1 <= c0) return 0;
  const unsigned grid = (unsigned)std::min<long>((rows * (c1 - c0) + 255) / 256, 1024);
  hipLaunchKernelGGL(checksum_cols_kernel, dim3(grid), dim3(256), 0, s, static_cast<const unsigned short*>(p), rows, ld, c0, c1, out);
  ROMA_LAUNCH_CHECK();
  return 0;
}

int checksum_launch(const void* p, size_t bytes, unsigned long long* out, hipStream_t s) {
  const long nwords = (long)(bytes / 4);
  if (nwords <= 0) return 0;
  const unsigned grid = (unsigned)std::min<long>((nwords + 255) / 256, 1024);
  hipLaunchKernelGGL(checksum_kernel, dim3(grid), dim3(256), 0, s, static_cast<const uint32_t*>(p), nwords, out);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
