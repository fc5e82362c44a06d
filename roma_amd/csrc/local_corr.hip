// Fused local-window correlation for gfx950 (see local_corr.h).
//
// HBM/LDS-bound gather + dot, NOT reshaped into a GEMM.  One wave64 per query pixel.
//
// Integer-patch identity: the (2r+1)^2 window taps are exactly one f1 pixel apart
// (romatch/utils/local_correlation.py:93-108), so every tap shares the same fractional
// offset (fx,fy) and
//     corr[j][i] = (1-fy)(1-fx) D[j][i] + (1-fy)fx D[j][i+1] + fy(1-fx) D[j+1][i] + fy fx D[j+1][i+1]
// with D[a][b] = <f0, f1[y0-r+a][x0-r+b]>, a,b in [0, 2r+2): (2r+2)^2 dot products instead of
// 4(2r+1)^2 bilinear taps, and every f1 row of the patch is ONE contiguous (2r+2)*C run in
// channels-last memory.
//
// Lane mapping: lane = pos*S + s; `pos` walks the 2r+2 patch columns, the S lanes of a column
// split the channels in interleaved 16-byte chunks, so one wave load instruction touches
// (2r+2) fully used 64..128-byte segments.  f0 is staged once per wave in LDS as f32.
#include "local_corr.h"
#include <stdio.h>
#include "gemm.h"  // DT_*

namespace roma {

template <typename T> struct LcIO;
template <> struct LcIO<float> {
  static constexpr int CE = 4;
  __device__ static inline void ld(const float* p, float* v) {
    f32x4 x = *reinterpret_cast<const f32x4*>(p);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  }
};
template <> struct LcIO<bf16_t> {
  static constexpr int CE = 8;
  __device__ static inline void ld(const bf16_t* p, float* v) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
  }
};

__device__ inline void unnormalize_floor(float w, int size, int& i0, float& frac) {
  // grid_sample, align_corners=False: ((x + 1) * size - 1) / 2
  float ix = ((w + 1.f) * size - 1.f) * 0.5f;
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);  // also maps NaN to a finite (all-zero-padding) location
  const float f = floorf(ix);
  i0 = (int)f;
  frac = ix - f;
}

template <int R, typename T, typename TOUT>
__global__ __launch_bounds__(256) void local_corr_window_kernel(const LocalCorrArgs a) {
  constexpr int P = 2 * R + 2;
  constexpr int PP = (P <= 8) ? 8 : 16;
  constexpr int S = 64 / PP;
  constexpr int CE = LcIO<T>::CE;
  constexpr int KW = 2 * R + 1;
  extern __shared__ __attribute__((aligned(16))) float f0s[];  // [4 waves][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long HW = (long)a.H * a.W;
  const long total = (long)a.B * HW;
  long pix = (long)blockIdx.x * 4 + wave;
  const bool active = pix < total;
  if (!active) pix = total - 1;
  const int b = (int)(pix / HW);
  const long p = pix - (long)b * HW;
  const int pos = lane / S, s = lane % S;

  // ---- stage f0 row (as f32) into this wave's LDS slice
  const T* f0p = reinterpret_cast<const T*>(a.f0) + ((long)b * HW + p) * a.ld0;
  float* myf0 = f0s + wave * a.C;
  for (int c = lane * CE; c < a.C; c += 64 * CE) {
    float v[CE];
    LcIO<T>::ld(f0p + c, v);
#pragma unroll
    for (int j = 0; j < CE; ++j) myf0[c + j] = v[j];
  }
  __syncthreads();

  int x0, y0;
  float fx, fy;
  unnormalize_floor(a.warp[pix * 2 + 0], a.W, x0, fx);
  unnormalize_floor(a.warp[pix * 2 + 1], a.H, y0, fy);
  const int x = x0 - R + pos;
  const bool xok = pos < P && x >= 0 && x < a.W;
  const int simg = (b + a.f1_shift) % a.nimg;
  const T* f1p = reinterpret_cast<const T*>(a.f1) + (long)simg * HW * a.ld1;
  const int NI = a.C / (CE * S);

  float D[P];
#pragma unroll
  for (int r = 0; r < P; ++r) {
    const int y = y0 - R + r;
    float sum = 0.f;
    if (y >= 0 && y < a.H && xok) {
      const T* src = f1p + ((long)y * a.W + x) * a.ld1 + CE * s;
      const float* fq = myf0 + CE * s;
#pragma unroll 4
      for (int i = 0; i < NI; ++i) {
        float v[CE];
        LcIO<T>::ld(src + (long)i * CE * S, v);
#pragma unroll
        for (int j = 0; j < CE; ++j) sum = fmaf(v[j], fq[i * CE * S + j], sum);
      }
    }
    D[r] = sum;
  }
  // ---- reduce the S channel slices of each column, fetch the right-hand neighbour column
  float D1[P];
#pragma unroll
  for (int r = 0; r < P; ++r) {
    float v = D[r];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    if (S == 8) v += __shfl_xor(v, 4);
    D[r] = v;
    D1[r] = __shfl_down(v, S);
  }
  if (active && pos < KW) {
    TOUT* o = reinterpret_cast<TOUT*>(a.out) + pix * a.ldo;
    const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      if ((j % S) == s) {
        const float c = w00 * D[j] + w01 * D1[j] + w10 * D[j + 1] + w11 * D1[j + 1];
        ElemIO<TOUT>::st(o + j * KW + pos, c * a.scale);
      }
    }
  }
}

// General per-tap form: warp[B,HW,K,2] arbitrary coordinates (plugin signature).
template <typename T, typename TOUT>
__global__ __launch_bounds__(256) void local_corr_general_kernel(const LocalCorrArgs a) {
  constexpr int CE = LcIO<T>::CE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long HW = (long)a.H * a.W;
  const long total = (long)a.B * HW;
  const long pix = (long)blockIdx.x * 4 + wave;
  if (pix >= total) return;
  const int b = (int)(pix / HW);
  const long p = pix - (long)b * HW;
  const T* f0p = reinterpret_cast<const T*>(a.f0) + ((long)b * HW + p) * a.ld0;
  const int simg = (b + a.f1_shift) % a.nimg;
  const T* f1p = reinterpret_cast<const T*>(a.f1) + (long)simg * HW * a.ld1;
  TOUT* o = reinterpret_cast<TOUT*>(a.out) + pix * a.ldo;
  for (int k = 0; k < a.K; ++k) {
    int x0, y0;
    float fx, fy;
    unnormalize_floor(a.warp[(pix * a.K + k) * 2 + 0], a.W, x0, fx);
    unnormalize_floor(a.warp[(pix * a.K + k) * 2 + 1], a.H, y0, fy);
    const float wgt[4] = {(1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx};
    float sum = 0.f;
    for (int c = lane * CE; c < a.C; c += 64 * CE) {
      float q[CE];
      LcIO<T>::ld(f0p + c, q);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
          float v[CE];
          LcIO<T>::ld(f1p + ((long)yy * a.W + xx) * a.ld1 + c, v);
          float d = 0.f;
#pragma unroll
          for (int j = 0; j < CE; ++j) d = fmaf(v[j], q[j], d);
          sum = fmaf(wgt[t], d, sum);
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) ElemIO<TOUT>::st(o + k, sum * a.scale);
  }
}

static int check_common(const LocalCorrArgs& a, int ce) {
  ROMA_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0, "local_corr: empty problem");
  ROMA_REQUIRE(a.ld0 % ce == 0 && a.ld1 % ce == 0, "local_corr: feature strides must keep 16-byte alignment");
  ROMA_REQUIRE((reinterpret_cast<uintptr_t>(a.f0) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.f1) & 15) == 0,
               "local_corr: features must be 16-byte aligned");
  ROMA_REQUIRE(a.nimg > 0, "local_corr: nimg must be positive");
  return 0;
}

template <int R>
static int launch_window_r(const LocalCorrArgs& a, hipStream_t stream) {
  const long total = (long)a.B * a.H * a.W;
  dim3 grid((unsigned)((total + 3) / 4));
  size_t lds = (size_t)4 * a.C * sizeof(float);
  // algorithmic bytes: f0 + f1 read once, centre warp, K outputs
  const double es_in = a.in_dt == DT_F32 ? 4.0 : 2.0, es_out = a.out_dt == DT_F32 ? 4.0 : 2.0;
  char pname[64];
  snprintf(pname, sizeof pname, "local_corr_window_kernel<%d,%s>", R, a.in_dt == DT_F32 ? "f32" : "bf16");
  ProfScope ps(pname, (double)total * (2.0 * a.C * es_in + 8.0 + (2.0 * R + 1) * (2.0 * R + 1) * es_out), "byte", stream);
#define ROMA_LC(T, TOUT) hipLaunchKernelGGL((local_corr_window_kernel<R, T, TOUT>), grid, dim3(256), lds, stream, a)
  if (a.in_dt == DT_F32 && a.out_dt == DT_F32) ROMA_LC(float, float);
  else if (a.in_dt == DT_F32) ROMA_LC(float, bf16_t);
  else if (a.out_dt == DT_F32) ROMA_LC(bf16_t, float);
  else ROMA_LC(bf16_t, bf16_t);
#undef ROMA_LC
  ROMA_LAUNCH_CHECK();
  return 0;
}

int local_corr_window_launch(const LocalCorrArgs& a, hipStream_t stream) {
  const int ce = a.in_dt == DT_F32 ? 4 : 8;
  if (int rc = check_common(a, ce)) return rc;
  const int P = 2 * a.radius + 2;
  const int S = 64 / (P <= 8 ? 8 : 16);
  ROMA_REQUIRE(a.C % (ce * S) == 0, "local_corr(window): C must be a multiple of 16B-chunk x lanes-per-column");
  ROMA_REQUIRE((size_t)4 * a.C * sizeof(float) <= 64 * 1024, "local_corr(window): C too large for the LDS f0 stage");
  switch (a.radius) {
    case 1: return launch_window_r<1>(a, stream);
    case 2: return launch_window_r<2>(a, stream);
    case 3: return launch_window_r<3>(a, stream);
    case 4: return launch_window_r<4>(a, stream);
    case 5: return launch_window_r<5>(a, stream);
    case 6: return launch_window_r<6>(a, stream);
    case 7: return launch_window_r<7>(a, stream);
    default: set_error("local_corr(window): radius must be in 1..7"); return -1;
  }
}

int local_corr_general_launch(const LocalCorrArgs& a, hipStream_t stream) {
  const int ce = a.in_dt == DT_F32 ? 4 : 8;
  if (int rc = check_common(a, ce)) return rc;
  ROMA_REQUIRE(a.C % ce == 0 && a.K > 0, "local_corr(general): bad C or K");
  const long total = (long)a.B * a.H * a.W;
  dim3 grid((unsigned)((total + 3) / 4));
#define ROMA_LC(T, TOUT) hipLaunchKernelGGL((local_corr_general_kernel<T, TOUT>), grid, dim3(256), 0, stream, a)
  if (a.in_dt == DT_F32 && a.out_dt == DT_F32) ROMA_LC(float, float);
  else if (a.in_dt == DT_F32) ROMA_LC(float, bf16_t);
  else if (a.out_dt == DT_F32) ROMA_LC(bf16_t, float);
  else ROMA_LC(bf16_t, bf16_t);
#undef ROMA_LC
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
