// 3x3 convolution (pad 1) + bias + ReLU for Cin = 64, Cout = 64 / 128, bf16 NHWC - VGG19 conv1_2 and conv2_1 of the match()
// path (encoders.py:17-27), the two layers with the most pixels (16 images x 560^2 / 864^2 and half that).
//
// Why not the implicit GEMM: at N = 64 / 128 a GEMM tile stages 256 x 64 (x 9 taps) of activations for every 64 / 128
// output columns - 51 / 85 FLOP per staged byte, feed bound at 0.36 / 0.60 PF (profiles/r02_final_*), and every input
// pixel crosses the L2 -> LDS path nine times.  Here the WEIGHTS never move: a wave keeps the 32 x 576 slice of W for its 32
// output channels in 144 VGPRs for its whole life, and the activations cross the LDS once: a workgroup walks down a strip
// of the image, each input row (+ 1 halo pixel per side) arrives ONCE in a 4-row LDS ring by LDS-DMA (pixel-swizzled 16-byte
// chunks, zero source for everything outside the image) and serves the 3 output rows and 9 taps that touch it.
//
//   * 4 waves per workgroup, two workgroups per CU.  Cout = 64: 2 channel halves x 2 pixel halves, 128 output pixels per
//     row; Cout = 128: 4 channel quarters, 64 output pixels per row.  Wave tile: 32 channels x 64 pixels = TM 2 blocks of
//     v_mfma_f32_32x32x16_bf16 with D[channel][pixel] (a lane ends up with 4 consecutive channels of one pixel).
//   * per output row: wait for input row y + 1 (issued one row earlier), barrier, issue row y + 2 into the slot row y - 2
//     left, then 9 taps x (8 fragment reads, wait, 8 MFMAs); the reads of tap t + 1 are issued right behind the MFMAs of
//     tap t.  Ring reads are inline asm (hipcc would drain the DMA queue before any LDS read that may alias a DMA target).
//   * LDS position (pixel p, chunk c) of a ring row holds source chunk c ^ (p & 7): the 32 pixels of a fragment read are
//     128 bytes apart, the swizzle spreads them over all banks.
//   * output: bias + ReLU + bf16 pack in registers, 8-byte stores (4 channels of one pixel per lane).
#include "conv64.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gemm_device.h"

namespace roma {

static __device__ __attribute__((aligned(256))) unsigned int g_c64_zero[64];   // zero source of out-of-image pixels
static __device__ __attribute__((aligned(256))) unsigned int g_c64_dump[128];  // where lanes right of the image store

int g_conv64_mode = -1;

#define C64_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

template <int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ out, int B,
                                                             int H, int W, int SY, int nxt, int nblocks) {
  constexpr int NCG = COUT / 32;        // channel groups (waves along channels)
  constexpr int NPG = 4 / NCG;          // pixel groups
  constexpr int TW = 64 * NPG;          // output pixels per row per workgroup
  constexpr int NPIECE = ((TW + 2) * 128 + 1023) / 1024;  // 1 KiB DMA pieces per ring row (17 ; 9)
  constexpr int RSTRIDE = NPIECE * 1024;
  constexpr int KW = (NPIECE + 3) / 4;  // pieces of the busiest wave (5 ; 3)
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];  // [4][RSTRIDE]

  const long lb = blockIdx.x;
  if (lb >= nblocks) return;
  const int xt = (int)(lb % nxt);
  long r = lb / nxt;
  const int yt = (H + SY - 1) / SY;
  const int ys = (int)(r % yt) * SY;
  const int b = (int)(r / yt);
  const int x0 = xt * TW;
  const int sy = min(SY, H - ys);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % NCG, pg = wave / NCG;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- this wave's weights: W[32 cg + l31][k = 16 ks + 8 h .. + 8), ks = tap * 4 + g  -> 36 x 16 bytes per lane
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 wreg[36];
  {
    const bf16_t* wp = w + (long)(32 * cg + l31) * 576 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 36; ++ks) wreg[ks] = *reinterpret_cast<const u32x4*>(wp + 16 * ks);
  }
  // bias lives in LDS behind the ring (registers are for W): channels 32 cg + 8 rg + 4 h + [0, 4) are read per output row
  float* bias_s = reinterpret_cast<float*>(ring + 4 * RSTRIDE);
  if (tid < COUT) bias_s[tid] = bias[tid];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the ordinary loads are retired before the first DMA is counted

  // ---- DMA of one input row (image row yy, may be outside the image) into ring slot `slot`: pieces wave, wave + 4, ...
  const char* zsrc = reinterpret_cast<const char*>(g_c64_zero);
  const bf16_t* inb = in + (long)b * H * W * 64;
  const int dl_p = lane >> 3, dl_slot = lane & 7;  // lane -> (pixel of the piece, 16-byte slot)
#define C64_ISSUE_ROW(YY, SLOT)                                                                                \
  {                                                                                                            \
    const int yy_ = (YY);                                                                                      \
    const bool rok_ = yy_ >= 0 && yy_ < H;                                                                     \
    _Pragma("unroll") for (int j = 0; j < KW; ++j) {                                                           \
      const int q_ = wave + 4 * j;                                                                             \
      if (q_ < NPIECE) {                                                                                       \
        const int p_ = q_ * 8 + dl_p;           /* ring pixel 0 .. TW + 1 (beyond: padding of the last piece) */ \
        const int x_ = x0 - 1 + p_;                                                                            \
        const int c_ = dl_slot ^ (p_ & 7);                                                                     \
        const bool ok_ = rok_ && p_ < TW + 2 && x_ >= 0 && x_ < W;                                             \
        const char* s_ = ok_ ? reinterpret_cast<const char*>(inb + ((long)yy_ * W + x_) * 64 + c_ * 8) : zsrc + dl_slot * 16; \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,                    \
                                         (__attribute__((address_space(3))) void*)(ring + (SLOT) * RSTRIDE + q_ * 1024), 16, 0, 0); \
      }                                                                                                        \
    }                                                                                                          \
  }

  // ---- fragment read offsets inside a ring row: pixel p = 64 pg + l31 + dx (+ 32 tm), chunk 2 g + h
  const unsigned ring0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)ring);
  // offset of chunk (2 g + h) ^ (p & 7) = (chunk h ^ (p & 7)) with bits 5-6 flipped by g: one register per dx
  unsigned rdo[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int p = 64 * pg + l31 + dx;
    rdo[dx] = ring0 + (unsigned)(p * 128 + ((h ^ (p & 7)) << 4));
  }
  u32x4 fa[2][4];  // [tm][g] fragments of the current tap
#define C64_READ_TAP(SLOTOFF, DX)                                                                             \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                             \
    const unsigned ad_ = (rdo[DX] + (SLOTOFF)) ^ (unsigned)(g << 5);                                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0][g]) : "v"(ad_));                                           \
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fa[1][g]) : "v"(ad_));                              \
  }
#define C64_WAIT_FRAGS()                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]), \
                 "+v"(fa[1][2]), "+v"(fa[1][3])::"memory")

  // ---- prologue: input rows ys - 1, ys, ys + 1 (ring slot = (row - ys + 1) & 3)
  C64_ISSUE_ROW(ys - 1, 0)
  C64_ISSUE_ROW(ys, 1)
  C64_ISSUE_ROW(ys + 1, 2)

  bf16_t* outb = out + (long)b * H * W * COUT;
  for (int o = 0; o < sy; ++o) {
    // input row ys + o + 1 (the last of this output row's three) was issued one iteration ago, behind it only the 8
    // output stores of that iteration; first iteration: everything the prologue issued
    if (o == 0) {
      C64_WAIT_VM(0);
    } else {
      C64_WAIT_VM(8);
    }
    __builtin_amdgcn_s_barrier();
    C64_ISSUE_ROW(ys + o + 2, (o + 3) & 3)  // into the slot of input row ys + o - 2: every wave is past its last read

    f32x16 acc[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][e] = 0.f;
    {
      const unsigned s0 = ((o + 0) & 3) * RSTRIDE;
      C64_READ_TAP(s0, 0)
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      C64_WAIT_FRAGS();
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wreg[t * 4 + g]),
                                                            __builtin_bit_cast(bf16x8_t, fa[tm][g]), acc[tm], 0, 0, 0);
      // the next tap's reads overwrite fa behind the ISSUED MFMAs (operands are read at issue, LDS data returns >= 64 cycles later)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < 9) {
        const unsigned sn = ((o + (t + 1) / 3) & 3) * RSTRIDE;
        C64_READ_TAP(sn, (t + 1) % 3)
      }
    }

    // ---- bias + ReLU + bf16, 8-byte stores: lane = (pixel l31 of block tm, channels 8 rg + 4 h + [0, 4))
    const int y = ys + o;
    f32x4 bv[4];
    {
      const unsigned ba = ring0 + 4 * RSTRIDE + (32 * cg + 4 * h) * 4;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                   "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]), "=&v"(bv[3])
                   : "v"(ba)
                   : "memory");
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int x = x0 + 64 * pg + 32 * tm + l31;
      bf16_t* op = outb + ((long)y * W + x) * COUT + 32 * cg + 4 * h;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        uint2 pk;
        pk.x = pack_bf16x2(fmaxf(acc[tm][4 * rg + 0] + bv[rg][0], 0.f), fmaxf(acc[tm][4 * rg + 1] + bv[rg][1], 0.f));
        pk.y = pack_bf16x2(fmaxf(acc[tm][4 * rg + 2] + bv[rg][2], 0.f), fmaxf(acc[tm][4 * rg + 3] + bv[rg][3], 0.f));
        // exactly 8 store instructions per wave and row, whatever the lanes' validity (the vmcnt above counts on it):
        // lanes right of the image store into a dump buffer
        uint2* dst = x < W ? reinterpret_cast<uint2*>(op + 8 * rg) : reinterpret_cast<uint2*>(g_c64_dump) + lane;
        *dst = pk;
      }
    }
  }
  C64_WAIT_VM(0);  // trailing DMA must not outlive the workgroup's LDS
#undef C64_WAIT_FRAGS
#undef C64_READ_TAP
#undef C64_ISSUE_ROW
}

// 0 = launched, 1 = not this kernel's problem
int conv64_try_launch(const GemmArgs& a, hipStream_t stream) {
  static const int use_env = getenv("ROMA_CONV64") ? atoi(getenv("ROMA_CONV64")) : 1;
  if (!(g_conv64_mode >= 0 ? g_conv64_mode : use_env)) return 1;
  if (a.conv_c != 64 || a.in_dt != DT_BF16 || a.out_dt != DT_BF16 || a.act != ACT_RELU || !a.bias) return 1;
  if ((a.N != 64 && a.N != 128) || a.ldc != a.N || a.ldw != 576 || a.batch != 1 || a.mode != EPI_STD || a.scale || a.res ||
      a.res_bf16 || a.alpha != 1.0f)
    return 1;
  const int H = a.conv_h, W = a.conv_w;
  const long hw = (long)H * W;
  if (hw <= 0 || a.M % hw != 0) return 1;
  const int B = (int)(a.M / hw);
  const int TW = a.N == 64 ? 128 : 64;
  const int SY = H >= 128 ? 32 : 16;
  const int nxt = (W + TW - 1) / TW;
  const long nb = (long)B * ((H + SY - 1) / SY) * nxt;
  if (nb <= 0 || nb >= (1l << 30)) return 1;
  char pname[64];
  snprintf(pname, sizeof pname, "conv3x3_c64_kernel<%d>", a.N);
  ProfScope ps(pname, 2.0 * (double)a.M * a.N * 576.0, "flop", stream);
  const bf16_t* in = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* w = reinterpret_cast<const bf16_t*>(a.W);
  bf16_t* out = reinterpret_cast<bf16_t*>(a.C);
  const size_t lds = (size_t)4 * ((((TW + 2) * 128 + 1023) / 1024) * 1024) + 512;  // ring + bias
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 17 * 1024 + 512));
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 9 * 1024 + 512));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  if (a.N == 64) {
    hipLaunchKernelGGL(conv3x3_c64_kernel<64>, dim3((unsigned)nb), dim3(256), lds, stream, in, w, a.bias, out, B, H, W, SY, nxt, (int)nb);
  } else {
    hipLaunchKernelGGL(conv3x3_c64_kernel<128>, dim3((unsigned)nb), dim3(256), lds, stream, in, w, a.bias, out, B, H, W, SY, nxt, (int)nb);
  }
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
