// Keypoint matching helpers (see keypoints.h).  All-pairs work of at most ~1e8 pairs: VALU bound, microseconds.
#include "keypoints.h"

#include <algorithm>

namespace roma {

// ------------------------------------------------------------------ bilinear sampling of (warp_B, certainty) at points
__global__ __launch_bounds__(256) void sample_warp_at_kernel(const float* __restrict__ warp, const float* __restrict__ cert,
                                                             int H, int W, const float* __restrict__ xa, long n,
                                                             float* __restrict__ xa_to_b, float* __restrict__ cert_a) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gx = xa[i * 2 + 0], gy = xa[i * 2 + 1];
  float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;  // align_corners=False
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
  iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float tx = ix - fx0, ty = iy - fy0;
  const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
  float bx = 0.f, by = 0.f, c = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {  // zeros padding
      const long p = (long)yy * W + xx;
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(warp + p * 4);
      bx += wgt[t] * w4[2];
      by += wgt[t] * w4[3];
      c += wgt[t] * cert[p];
    }
  }
  xa_to_b[i * 2 + 0] = bx;
  xa_to_b[i * 2 + 1] = by;
  cert_a[i] = c;
}

int sample_warp_at_launch(const float* warp, const float* cert, int H, int W, const float* xa, long n, float* xa_to_b,
                          float* cert_a, hipStream_t s) {
  if (n == 0) return 0;
  ROMA_REQUIRE(warp && cert && xa && xa_to_b && cert_a && H > 0 && W > 0 && n > 0, "sample_warp_at: bad arguments");
  ROMA_REQUIRE((reinterpret_cast<uintptr_t>(warp) & 15) == 0, "sample_warp_at: warp must be 16-byte aligned [H,W,4] f32");
  if (n == 0) return 0;
  hipLaunchKernelGGL(sample_warp_at_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, warp, cert, H, W, xa, n,
                     xa_to_b, cert_a);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ nearest neighbour of every p_i in q (squared distance)
// best[i] = (float bits of min d2) << 32 | argmin j  (d2 >= 0, so the integer order is the numeric order and ties
// resolve to the lowest j); slices of q are merged with a 64-bit atomicMin.
__global__ __launch_bounds__(256) void nn_kernel(const float* __restrict__ p, long np, const float* __restrict__ q, long nq,
                                                 unsigned long long* __restrict__ best, long q_per_slice) {
  __shared__ float qs[1024 * 2];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float px = 0.f, py = 0.f;
  if (i < np) {
    px = p[i * 2 + 0];
    py = p[i * 2 + 1];
  }
  const long j_begin = (long)blockIdx.y * q_per_slice;
  const long j_end = min(nq, j_begin + q_per_slice);
  float bd = INFINITY;
  unsigned bj = 0xffffffffu;
  for (long j0 = j_begin; j0 < j_end; j0 += 1024) {
    const int cnt = (int)min((long)1024, j_end - j0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 2; t += 256) qs[t] = q[j0 * 2 + t];
    __syncthreads();
    for (int t = 0; t < cnt; ++t) {
      const float dx = px - qs[2 * t], dy = py - qs[2 * t + 1];
      const float d2 = fmaf(dy, dy, dx * dx);
      if (d2 < bd) {  // strict: the first (lowest) index wins ties
        bd = d2;
        bj = (unsigned)(j0 + t);
      }
    }
  }
  if (i < np && bj != 0xffffffffu)
    atomicMin(best + i, ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)bj);
}

__global__ __launch_bounds__(256) void mutual_nn_finalize_kernel(const unsigned long long* __restrict__ best_a,
                                                                 const unsigned long long* __restrict__ best_b, long na,
                                                                 const float* __restrict__ cert_a, float cert_th,
                                                                 float max_d2, int* __restrict__ match_b) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= na) return;
  const unsigned long long ba = best_a[i];
  const unsigned j = (unsigned)(ba & 0xffffffffull);
  const unsigned d2bits = (unsigned)(ba >> 32);
  int out = -1;
  if (j != 0xffffffffu) {
    // (D == row min) * (D == column min) * (certainty > th) * (D < max_dist)   (matcher.py:756-762)
    const bool col_min = (unsigned)(best_b[j] >> 32) == d2bits;
    const bool cert_ok = cert_a == nullptr || cert_a[i] > cert_th;
    if (col_min && cert_ok && __uint_as_float(d2bits) < max_d2) out = (int)j;
  }
  match_b[i] = out;
}

int mutual_nn_launch(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                     int* match_b, unsigned long long* ws_a, unsigned long long* ws_b, hipStream_t s) {
  if (na == 0) return 0;
  ROMA_REQUIRE(a && match_b && ws_a && ws_b && na > 0 && nb >= 0 && (b || nb == 0), "mutual_nn: bad arguments");
  ROMA_REQUIRE(na < (1l << 31) && nb < (1l << 31), "mutual_nn: too many points");
  if (na == 0) return 0;
  ROMA_CHECK_HIP(hipMemsetAsync(ws_a, 0xff, (size_t)na * 8, s));
  if (nb > 0) ROMA_CHECK_HIP(hipMemsetAsync(ws_b, 0xff, (size_t)nb * 8, s));
  auto run = [&](const float* p, long np, const float* q, long nq, unsigned long long* best) {
    const long pblocks = (np + 255) / 256;
    long slices = std::max<long>(1, std::min<long>((nq + 1023) / 1024, (1024 + pblocks - 1) / pblocks));
    const long per = (((nq + slices - 1) / slices) + 1023) / 1024 * 1024;
    slices = (nq + per - 1) / per;
    hipLaunchKernelGGL(nn_kernel, dim3((unsigned)pblocks, (unsigned)slices), dim3(256), 0, s, p, np, q, nq, best, per);
  };
  if (nb > 0) {
    run(a, na, b, nb, ws_a);
    run(b, nb, a, na, ws_b);
  }
  hipLaunchKernelGGL(mutual_nn_finalize_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, ws_a, ws_b, na, cert_a,
                     cert_th, max_dist * max_dist, match_b);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ tie-complete mutual matches
// The reference returns torch.nonzero of the full mask (D == row min) * (D == column min) * (cert > th) * (D < max_dist)
// (matcher.py:756-762): EVERY tied pair, row-major.  Ties are what duplicate keypoints produce (detectors emit the same
// location at several scales), so they are exact in any arithmetic.  After the two nearest-neighbour passes the row / column
// minima are known; a row's matches are then all j whose squared distance has the bits of BOTH minima.
//   MODE 0: offs[i + 1] = number of matches of row i (offs[0] = 0); an in-place inclusive scan follows
//   MODE 1: pairs[offs[i] + k] = (i, j_k), j ascending
template <int MODE>
__global__ __launch_bounds__(256) void mutual_pairs_kernel(const float* __restrict__ a, long na, const float* __restrict__ b, long nb,
                                                           const unsigned long long* __restrict__ best_a,
                                                           const unsigned long long* __restrict__ best_b,
                                                           const float* __restrict__ cert_a, float cert_th, float max_d2,
                                                           long long* __restrict__ offs, long long* __restrict__ pairs) {
  __shared__ float qs[1024 * 2];
  __shared__ unsigned qmin[1024];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float px = 0.f, py = 0.f;
  unsigned rowmin = 0xffffffffu;
  bool live = false;
  if (i < na) {
    px = a[i * 2 + 0];
    py = a[i * 2 + 1];
    const unsigned long long ba = best_a[i];
    rowmin = (unsigned)(ba >> 32);
    live = (unsigned)(ba & 0xffffffffull) != 0xffffffffu && (cert_a == nullptr || cert_a[i] > cert_th) &&
           __uint_as_float(rowmin) < max_d2;
  }
  long n = 0;
  const long base = (MODE == 1 && i < na) ? offs[i] : 0;
  for (long j0 = 0; j0 < nb; j0 += 1024) {
    const int cnt = (int)min((long)1024, nb - j0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 2; t += 256) qs[t] = b[j0 * 2 + t];
    for (int t = threadIdx.x; t < cnt; t += 256) qmin[t] = (unsigned)(best_b[j0 + t] >> 32);
    __syncthreads();
    if (!live) continue;
    for (int t = 0; t < cnt; ++t) {
      if (qmin[t] != rowmin) continue;
      const float dx = px - qs[2 * t], dy = py - qs[2 * t + 1];
      if (__float_as_uint(fmaf(dy, dy, dx * dx)) != rowmin) continue;  // same expression as nn_kernel
      if (MODE == 1) {
        pairs[(base + n) * 2 + 0] = i;
        pairs[(base + n) * 2 + 1] = j0 + t;
      }
      ++n;
    }
  }
  if (MODE == 0 && i < na) offs[i + 1] = n;
}

// in-place inclusive scan of v[1 .. n] (v[0] = 0): one workgroup, 1024-element chunks
__global__ __launch_bounds__(1024) void scan_inclusive_kernel(long long* __restrict__ v, long n) {
  __shared__ long long part[1024];
  __shared__ long long carry;
  if (threadIdx.x == 0) {
    carry = 0;
    v[0] = 0;
  }
  __syncthreads();
  for (long c0 = 1; c0 <= n; c0 += 1024) {
    const long idx = c0 + threadIdx.x;
    part[threadIdx.x] = idx <= n ? v[idx] : 0;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const long long add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (idx <= n) v[idx] = part[threadIdx.x] + carry;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
}

int mutual_nn_count_launch(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                           unsigned long long* ws_a, unsigned long long* ws_b, long long* offs, hipStream_t s) {
  ROMA_REQUIRE(offs && na >= 0, "mutual_nn_count: bad arguments");
  if (na == 0) {
    ROMA_CHECK_HIP(hipMemsetAsync(offs, 0, sizeof(long long), s));
    return 0;
  }
  ROMA_REQUIRE(a && ws_a && ws_b && nb >= 0 && (b || nb == 0), "mutual_nn_count: bad arguments");
  ROMA_REQUIRE(na < (1l << 31) && nb < (1l << 31), "mutual_nn_count: too many points");
  ROMA_CHECK_HIP(hipMemsetAsync(ws_a, 0xff, (size_t)na * 8, s));
  if (nb > 0) ROMA_CHECK_HIP(hipMemsetAsync(ws_b, 0xff, (size_t)nb * 8, s));
  auto run = [&](const float* p, long np, const float* q, long nq, unsigned long long* best) {
    const long pblocks = (np + 255) / 256;
    long slices = std::max<long>(1, std::min<long>((nq + 1023) / 1024, (1024 + pblocks - 1) / pblocks));
    const long per = (((nq + slices - 1) / slices) + 1023) / 1024 * 1024;
    slices = (nq + per - 1) / per;
    hipLaunchKernelGGL(nn_kernel, dim3((unsigned)pblocks, (unsigned)slices), dim3(256), 0, s, p, np, q, nq, best, per);
  };
  if (nb > 0) {
    run(a, na, b, nb, ws_a);
    run(b, nb, a, na, ws_b);
  }
  hipLaunchKernelGGL(mutual_pairs_kernel<0>, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, a, na, b, nb, ws_a, ws_b, cert_a,
                     cert_th, max_dist * max_dist, offs, (long long*)nullptr);
  hipLaunchKernelGGL(scan_inclusive_kernel, dim3(1), dim3(1024), 0, s, offs, na);
  ROMA_LAUNCH_CHECK();
  return 0;
}

int mutual_nn_fill_launch(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                          const unsigned long long* ws_a, const unsigned long long* ws_b, long long* offs, long long* pairs,
                          hipStream_t s) {
  if (na == 0 || nb == 0) return 0;
  ROMA_REQUIRE(a && b && ws_a && ws_b && offs && pairs, "mutual_nn_fill: bad arguments");
  hipLaunchKernelGGL(mutual_pairs_kernel<1>, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, a, na, b, nb, ws_a, ws_b, cert_a,
                     cert_th, max_dist * max_dist, offs, pairs);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ forward-backward consistency (matcher.py:672-699)
// in_th[b,y,x] = || grid(x,y) - bilinear_zeropad(flow_backward[b], flow_forward[b,y,x]) || < th_n
__global__ __launch_bounds__(256) void fb_consistency_kernel(const float* __restrict__ ff, const float* __restrict__ fb,
                                                             int B, int H, int W, float th_n, float* __restrict__ out) {
  const long total = (long)B * H * W;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  const long r = idx / W;
  const int y = (int)(r % H);
  const long b = r / H;
  const float gx = ff[idx * 2 + 0], gy = ff[idx * 2 + 1];
  float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
  iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float tx = ix - fx0, ty = iy - fy0;
  const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
  float cx = 0.f, cy = 0.f;
  const float* fbb = fb + b * (long)H * W * 2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const long p = ((long)yy * W + xx) * 2;
      cx += wgt[t] * fbb[p];
      cy += wgt[t] * fbb[p + 1];
    }
  }
  const float dx = (-1.f + (2.f * x + 1.f) / W) - cx, dy = (-1.f + (2.f * y + 1.f) / H) - cy;
  out[idx] = sqrtf(dx * dx + dy * dy) < th_n ? 1.f : 0.f;
}

int fb_consistency_launch(const float* flow_fwd, const float* flow_bwd, int B, int H, int W, float th_n, float* out,
                          hipStream_t s) {
  ROMA_REQUIRE(flow_fwd && flow_bwd && out && B > 0 && H > 0 && W > 0, "fb_consistency: bad arguments");
  const long total = (long)B * H * W;
  hipLaunchKernelGGL(fb_consistency_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, flow_fwd, flow_bwd, B, H, W,
                     th_n, out);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ visualize_warp (matcher.py:936-986)
// out[c, y, x] = cert[y, x] * bilinear_zeropad(src(x)[c], grid(y, x)) + (1 - cert[y, x])     (white background)
// left half (x < W): src = im_B sampled at warp[y, x, 2:4]; right half (symmetric only): src = im_A at warp[y, x, 0:2]
__global__ __launch_bounds__(256) void visualize_warp_kernel(const float* __restrict__ warp, const float* __restrict__ cert,
                                                             const float* __restrict__ im_a, const float* __restrict__ im_b,
                                                             int H, int W, int W2, int hi, int wi, float* __restrict__ out) {
  const long total = (long)H * W2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W2);
  const bool right = x >= W;
  const float* src = right ? im_a : im_b;
  const float gx = warp[idx * 4 + (right ? 0 : 2)], gy = warp[idx * 4 + (right ? 1 : 3)];
  float ix = ((gx + 1.f) * wi - 1.f) * 0.5f, iy = ((gy + 1.f) * hi - 1.f) * 0.5f;
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
  iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float tx = ix - fx0, ty = iy - fy0;
  const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
  float rgb[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
    if (yy >= 0 && yy < hi && xx >= 0 && xx < wi) {
      const long p = (long)yy * wi + xx;
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] += wgt[t] * src[(long)c * hi * wi + p];
    }
  }
  const float ce = cert[idx];
#pragma unroll
  for (int c = 0; c < 3; ++c) out[(long)c * total + idx] = ce * rgb[c] + (1.f - ce);
}

int visualize_warp_launch(const float* warp, const float* cert, const float* im_a, const float* im_b, int H, int W,
                          int symmetric, int im_h, int im_w, float* out, hipStream_t s) {
  ROMA_REQUIRE(warp && cert && im_b && out && H > 0 && W > 0 && im_h > 0 && im_w > 0, "visualize_warp: bad arguments");
  ROMA_REQUIRE(!symmetric || im_a, "visualize_warp: symmetric warps need im_A as well");
  const int W2 = symmetric ? 2 * W : W;
  const long total = (long)H * W2;
  hipLaunchKernelGGL(visualize_warp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, warp, cert, im_a, im_b, H, W,
                     W2, im_h, im_w, out);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
