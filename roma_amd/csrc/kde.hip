// All-pairs Gaussian KDE in 4-D (see kde.h).  VALU-bound: per pair 4 sub + 4 fma + 1 v_exp_f32 + 1 add.
//   * one thread per query point (registers), 256 queries per workgroup;
//   * reference points stream through LDS in tiles of 1024 (16 KB, float4 each): every LDS read is a wave-wide
//     broadcast of one 16-byte point, so the inner loop is 1 ds_read_b128 per 64 x 10 VALU operations;
//   * the reference set is split over gridDim.y slices (atomicAdd of the partial sums) so that n = 40 000 still
//     fills the 256 CUs (157 query blocks alone would leave 40 % of them idle).
#include "kde.h"

#include <algorithm>

namespace roma {

__device__ __forceinline__ float round_to_half(float v) { return (float)(_Float16)v; }

template <bool HALF>
__global__ __launch_bounds__(256) void kde_kernel(const float* __restrict__ x, long n, int down, long nref, float coef,
                                                  float* __restrict__ density, long ref_per_slice) {
  __shared__ __attribute__((aligned(16))) f32x4 ys[1024];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  f32x4 xi = {0.f, 0.f, 0.f, 0.f};
  if (i < n) xi = *reinterpret_cast<const f32x4*>(x + i * 4);
  if (HALF) {
#pragma unroll
    for (int k = 0; k < 4; ++k) xi[k] = round_to_half(xi[k]);
  }
  const long j_begin = (long)blockIdx.y * ref_per_slice;
  const long j_end = min(nref, j_begin + ref_per_slice);
  float acc = 0.f;
  for (long j0 = j_begin; j0 < j_end; j0 += 1024) {
    const int cnt = (int)min((long)1024, j_end - j0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      f32x4 y = *reinterpret_cast<const f32x4*>(x + (j0 + t) * (long)down * 4);
      if (HALF) {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = round_to_half(y[k]);
      }
      ys[t] = y;
    }
    __syncthreads();
#pragma unroll 8
    for (int t = 0; t < cnt; ++t) {
      const f32x4 y = ys[t];
      const float d0 = xi[0] - y[0], d1 = xi[1] - y[1], d2 = xi[2] - y[2], d3 = xi[3] - y[3];
      const float q = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
      acc += __builtin_amdgcn_exp2f(q * coef);  // coef = -log2(e) / (2 std^2)
    }
  }
  if (i < n) atomicAdd(density + i, acc);
}

int kde_launch(const float* x, long n, int down, float std, int half_inputs, float* density, hipStream_t s) {
  ROMA_REQUIRE(x && density && n > 0, "kde: empty input");
  ROMA_REQUIRE(down >= 1 && std > 0.f, "kde: down must be >= 1 and std > 0");
  ROMA_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "kde: points must be 16-byte aligned [n,4] f32");
  const long nref = (n + down - 1) / down;  // x[::down]
  const long qblocks = (n + 255) / 256;
  long slices = std::max<long>(1, std::min<long>((nref + 1023) / 1024, (4 * 256 + qblocks - 1) / qblocks));
  const long per = (((nref + slices - 1) / slices) + 1023) / 1024 * 1024;
  slices = (nref + per - 1) / per;
  ROMA_CHECK_HIP(hipMemsetAsync(density, 0, (size_t)n * sizeof(float), s));
  const float coef = -1.4426950408889634f / (2.0f * std * std);
  dim3 grid((unsigned)qblocks, (unsigned)slices);
  // algorithmic work: 10 FLOP-equivalents per pair (4 sub, 4 mul-add, scale, exp) - VALU bound
  ProfScope ps(half_inputs ? "kde_kernel<half>" : "kde_kernel<f32>", 10.0 * (double)n * (double)nref, "flop", s);
  if (half_inputs) hipLaunchKernelGGL(kde_kernel<true>, grid, dim3(256), 0, s, x, n, down, nref, coef, density, per);
  else hipLaunchKernelGGL(kde_kernel<false>, grid, dim3(256), 0, s, x, n, down, nref, coef, density, per);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
