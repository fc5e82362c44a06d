// MaxPool 2x2 + projection head of one VGG pyramid level in ONE pass over the level's feature map (16-bit modes).
//
// The reference keeps the un-pooled map of every VGG block as a pyramid feature (encoders.py:17-27: feats[scale] = x in front
// of each MaxPool2d) and sends it through decoder.proj[scale] = Conv2d(C, Cf, 1) + BatchNorm2d (roma_models.py:156-160) when the
// decoder reaches that scale.  Until round 5 that was two bandwidth-bound kernels per level that each read the whole map:
// maxpool_kernel (at its HBM roof) and the proj GEMM with N = 9 / 64 outputs (K = 64 / 128: ~9 / 42 FLOP per byte).  At stride 1
// the map is 1.53 GB per batch-8 upsample pass; reading it once instead of twice is worth ~0.4 ms, at stride 2 ~0.2 ms.
//
// One wave owns a 2-row x 32-pixel patch: lane (pixel l31, half h) loads the 16 bytes (8 channels) at channel 16 ks + 8 h of
// its pixel for both rows - exactly the B fragments of v_mfma_f32_32x32x16 (the proj GEMM's own operand layout, so the
// products are summed in the same order as gemm_kernel does: bit-identical outputs) - runs D[cout][pixel] += W . x with the
// folded weights held in registers, adds the bias, pairs the half-waves with v_permlane32_swap into 16-byte stores, and takes
// the 2x2 maximum of the SAME registers: vertical max in the lane, horizontal max with the neighbour lane, even lanes store.
// The maxima are taken on the packed 16-bit patterns as unsigned integers: the inputs come out of a ReLU (never negative,
// never -0), and for non-negative bfloat16 / binary16 values integer order is numeric order - the same result as
// fmaxf on the widened values (maxpool_kernel), bit for bit.
#include "pool_proj.h"

#include <algorithm>

#include "gemm.h"  // DT_*

namespace roma {

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u32x4 pk_max_u16(u32x4 a, u32x4 b) {
  return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(u16x8, a), __builtin_bit_cast(u16x8, b)));
}
}  // namespace

// CIN: channels of the map (64 / 128); NB: 32-row blocks of proj outputs (1: N <= 32, 2: N <= 64)
template <int CIN, int NB>
__global__ __launch_bounds__(256) void pool_proj_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ pooled,
                                                        bf16_t* __restrict__ pf, const bf16_t* __restrict__ pw, long ldw,
                                                        const float* __restrict__ pb, int N, int ldf, int nimg, int H, int W,
                                                        long ntasks) {
  constexpr int KS = CIN / 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const long task = (long)blockIdx.x * 4 + wave;
  if (task >= ntasks) return;
  const int nxt = (W + 31) / 32, nyp = (H + 1) / 2;
  const int xt = (int)(task % nxt);
  const long r = task / nxt;
  const int jp = (int)(r % nyp);
  const int b = (int)(r / nyp);
  const int Ho = H / 2, Wo = W / 2;

  // folded proj weights as A fragments: W[32 nb + l31][16 ks + 8 h .. + 8); rows >= N are zero (their outputs are the zero pad)
  u32x4 wf[NB][KS];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int j = 32 * nb + l31;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wf[nb][ks] = u32x4{0u, 0u, 0u, 0u};
      if (j < N) wf[nb][ks] = *reinterpret_cast<const u32x4*>(pw + (long)j * ldw + 16 * ks + 8 * h);
    }
  }
  const int x = xt * 32 + l31;
  const bool xok = x < W;
  const int xc = xok ? x : W - 1;  // clamped address, masked stores
  u32x4 xf[2][KS];
  const int y0 = 2 * jp;
  const bool two = y0 + 1 < H;
#pragma unroll
  for (int ry = 0; ry < 2; ++ry) {
    const int y = (ry == 1 && !two) ? y0 : y0 + ry;
    const bf16_t* p = in + (((long)b * H + y) * W + xc) * CIN + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ry][ks] = *reinterpret_cast<const u32x4*>(p + 16 * ks);
  }

  // ---- proj: D[cout][pixel] for both rows
#pragma unroll
  for (int ry = 0; ry < 2; ++ry) {
    if (ry == 1 && !two) break;
    const int y = y0 + ry;
    bf16_t* orow = pf + (((long)b * H + y) * W + x) * ldf;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = mfma_h16_32x32x16(wf[nb][ks], xf[ry][ks], acc);
      // lane (pixel l31, h) holds couts 32 nb + 8 g + 4 h + [0, 4), g = 0 .. 3; pairs of groups (P = g / 2) go through
      // v_permlane32_swap so that lane (l31, h) ends up with the 8 consecutive couts 32 nb + 16 P + 8 h + [0, 8)
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        const int c0 = 32 * nb + 16 * P;
        if (c0 >= ldf) break;  // (N = 9: ldf = 16 - only P = 0 of block 0 exists)
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int j = c0 + 8 * (i >> 2) + 4 * h + (i & 3);
          v[i] = acc[8 * P + i] + (j < N ? pb[j] : 0.f);
        }
        const unsigned a0 = pack_bf16x2(v[0], v[1]), a1 = pack_bf16x2(v[2], v[3]);
        const unsigned b0 = pack_bf16x2(v[4], v[5]), b1 = pack_bf16x2(v[6], v[7]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4 st = {s0[0], s1[0], s0[1], s1[1]};
        if (xok) *reinterpret_cast<u32x4*>(orow + c0 + 8 * h) = st;
      }
    }
  }

  // ---- 2 x 2 maximum: vertical in the lane, horizontal with lane + 1, even lanes store pooled pixel x / 2
  if (two && jp < Ho) {
    bf16_t* prow = pooled + (((long)b * Ho + jp) * Wo + (x >> 1)) * CIN + 8 * h;
    const bool st_ok = !(l31 & 1) && (x >> 1) < Wo && x + 1 < W;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 m = pk_max_u16(xf[0][ks], xf[1][ks]);
      u32x4 nbr;
#pragma unroll
      for (int i = 0; i < 4; ++i) nbr[i] = (unsigned)__shfl_down((int)m[i], 1);
      const u32x4 mm = pk_max_u16(m, nbr);
      if (st_ok) *reinterpret_cast<u32x4*>(prow + 16 * ks) = mm;
    }
  }
}

bool pool_proj_supported(int C, int N, int ldf, int dt) {
  return dt == DT_BF16 && ((C == 64 && N <= 32) || (C == 128 && N <= 64)) && ldf % 8 == 0 && ldf >= N && ldf <= (N <= 32 ? 32 : 64);
}

int pool_proj_launch(const void* in, void* pooled, void* pf, const void* pw, long ldw, const float* pb, int N, int ldf, int nimg,
                     int H, int W, int C, int dt, hipStream_t s) {
  ROMA_REQUIRE(pool_proj_supported(C, N, ldf, dt), "pool_proj: 16-bit maps with C = 64 (N <= 32) or C = 128 (N <= 64) only");
  ROMA_REQUIRE(in && pooled && pf && pw && pb && H >= 2 && W >= 2 && ldw % 8 == 0, "pool_proj: bad arguments");
  const long ntasks = (long)nimg * ((H + 1) / 2) * ((W + 31) / 32);
  ROMA_REQUIRE(ntasks > 0 && (ntasks + 3) / 4 < (1l << 31), "pool_proj: grid too large");
  const double es = 2.0;
  // algorithmic bytes: the map once, the pooled map, the projected map
  ProfScope ps(C == 64 ? "pool_proj_kernel<64>" : "pool_proj_kernel<128>",
               (double)nimg * ((double)H * W * C * es + (double)(H / 2) * (W / 2) * C * es + (double)H * W * N * es), "byte", s);
  const dim3 grid((unsigned)((ntasks + 3) / 4));
  if (C == 64)
    hipLaunchKernelGGL((pool_proj_kernel<64, 1>), grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)pooled, (bf16_t*)pf,
                       (const bf16_t*)pw, ldw, pb, N, ldf, nimg, H, W, ntasks);
  else
    hipLaunchKernelGGL((pool_proj_kernel<128, 2>), grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)pooled, (bf16_t*)pf,
                       (const bf16_t*)pw, ldw, pb, N, ldf, nimg, H, W, ntasks);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
