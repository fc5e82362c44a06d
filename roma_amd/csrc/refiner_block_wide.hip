// Fused ConvRefiner block for the WIDE scales (round 5): out = conv1x1( relu( bn( dwconv5x5(in) ) ) ) in ONE kernel, C = 576
// (stride 4, both passes: 18 of the 45 wide blocks of a call and 60 % of their bytes).  romatch/models/matcher.py:92-122, 175-176.
//
// Why.  As two kernels the block moves the activation tensor through HBM four times (dwconv5x5: read + write, 1x1 GEMM: read +
// write); the stand-alone stencil already runs at 88 % of a copy and the weight-stationary GEMM at the vendor library's rate, so
// neither gets faster by itself - the round trip of the intermediate has to go.  Round 3 costed the fusion as "the stencil as the
// A-operand producer of the 256 x 192 GEMM tile" and rejected it: an n-tile re-produces its A tile, i.e. the 25-tap VALU work
// is done N / 192 = 3 times.  This kernel avoids the redundancy the other way round: a workgroup owns ALL 576 output channels of
// its pixels, so the stencil of a pixel runs exactly once.
//
// Anatomy (gfx950, one 512-thread workgroup = 8 waves per CU, 2 waves per SIMD, 256 registers each):
//   * tile = 8 rows x 16 columns = 128 pixels x all 576 output channels: 128 x 576 f32 accumulators = 144 registers per lane,
//     as 9 x 4 blocks of v_mfma_f32_16x16x32 per wave (wave = 64 pixels x 144 output channels; 13 fragment reads per 36 MFMAs);
//   * K runs over the 576 input channels in 9 slabs of 64.  Per slab:
//       - the input patch of the slab, (8 + 4) x (16 + 4) pixels x 128 B, arrives by LDS-DMA one slab ahead (two buffers; image
//         borders read a zero page: the convolution's zero padding costs no VALU);
//       - STENCIL: lane = (column, channel pair); it walks the 12 patch rows once (5 x ds_read_b32 per row, each 32-lane half
//         reads one pixel's whole 128-byte line: conflict free), feeds the rolling accumulators of the <= 5 output rows a patch
//         row touches with v_pk_fma_f32 in dwconv5x5_kernel's order (bias first, taps in raster order: the depthwise result is
//         bit-identical to the stand-alone kernel's), applies ReLU, rounds to the 16-bit format and writes the [128 px][64 k]
//         A tile in the GEMM kernels' swizzled LDS layout (chunk ^ ((row >> 1) & 7)): the intermediate never leaves the CU;
//       - the 576 x 64 slab of the 1x1 weights (72 KiB, the same swizzle applied on the DMA's source address) lands in the
//         meantime; MFMA: 2 k-steps x 36 MFMAs per wave, weights and activations both from LDS;
//       - two workgroup barriers per slab (A tile complete / A tile and W slab consumed).
//     The tap weights of the next slab (25 x 2 f32 per lane) are fetched under the MFMA phase.
//   * epilogue: + bias, round, stage half a tile at a time in LDS ([64 px][1152 + 16 B]) and leave as 16-byte row segments.
//   LDS: 2 x 32 KiB patch + 72 KiB W slab (dynamic, the DMA target: read with inline asm) + 16 KiB A tile (static: ordinary
//   code, hipcc orders only may-alias LDS accesses behind in-flight LDS-DMA) = 152 KiB.
//
// Per pixel the kernel reads its input once (+ the halo, served by L2: 1.9 x per tile) and writes its output once: 2 x C x 2 B
// of HBM traffic instead of 4 x.  Work per slab and CU: 288 MFMA-32-cycle equivalents (2 304 cycles per SIMD) next to ~400
// v_pk_fma_f32 + ~330 other VALU per wave; v0 runs the two phases back to back (all waves in lock-step).
#include "refiner_block.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gemm.h"  // DT_*

namespace roma {

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int rw_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int rw_u32x2;
#define ROMA_LDS __attribute__((address_space(3)))
typedef ROMA_LDS unsigned char lds_u8;

constexpr int RW_C = 576;                     // channels (in = out)
constexpr int RW_NS = RW_C / 64;              // K slabs of 64 channels
constexpr int RW_TH = 8, RW_TW = 16;          // output tile: 8 rows x 16 columns
constexpr int RW_PW = RW_TW + 4;              // patch: 12 rows x 20 columns of 128-byte pixels
constexpr int RW_PATCH_B = 32 * 1024;         // one patch buffer: 240 pixels + 16 unused slots (32 DMA instructions of 1 KiB)
constexpr int RW_W_B = RW_C * 128;            // one W slab: 576 rows x 128 B = 72 KiB
constexpr int RW_DYN = 2 * RW_PATCH_B + RW_W_B;  // 139 264 B dynamic LDS
constexpr int RW_OPITCH = RW_C * 2 + 16;      // staged output pixel pitch (1168 B: 16-byte aligned, 2-way on the 8-byte writes)
static_assert(64 * RW_OPITCH <= RW_DYN, "half a tile of staged output fits the dynamic region");
static_assert(RW_DYN + 128 * 128 <= 160 * 1024, "LDS");

__device__ __forceinline__ void rw_glds16(const char* src, lds_u8* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (ROMA_LDS void*)lds_wave_base, 16, 0, 0);
}
#define ROMA_RW_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define ROMA_RW_BARRIER()                            \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_s_barrier();                      \
  asm volatile("" ::: "memory")

__global__ __launch_bounds__(512, 2) void refiner_block_wide_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                                    const float* __restrict__ dww, const float* __restrict__ dwb,
                                                                    const bf16_t* __restrict__ pw, long ldpw,
                                                                    const float* __restrict__ pwb, int B, int H, int W, int nty,
                                                                    int ntx, long ntiles) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char dyn[];  // [patch 0][patch 1][W slab]: DMA targets
  __shared__ __attribute__((aligned(1024))) unsigned char atile[128 * 128];  // [128 px][64 k] 16-bit, swizzled 16-byte chunks
  lds_u8* const P0 = (lds_u8*)dyn;
  lds_u8* const WB = (lds_u8*)dyn + 2 * RW_PATCH_B;
  lds_u8* const AT = (lds_u8*)atile;

  // each XCD owns a contiguous band of tiles (the halo of a tile is its neighbours' interior: served by the XCD's own L2)
  const long per_xcd = (ntiles + 7) / 8;
  const long lt = (long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lt >= ntiles) return;
  const int tx = (int)(lt % ntx);
  long r_ = lt / ntx;
  const int ty = (int)(r_ % nty);
  const int b = (int)(r_ / nty);
  const int y0 = ty * RW_TH, x0 = tx * RW_TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---------------------------------------------------------------- DMA descriptors
  // Every source address is a wave-uniform base + a 32-bit per-lane offset (the saddr form of global_load_lds: no 64-bit
  // pointer registers - the first build kept four of them in scratch, and a scratch reload is a VMEM operation inside the
  // counted DMA stream).
  const char* const imb = reinterpret_cast<const char*>(in + (long)b * H * W * RW_C);
  // patch: instruction i of wave wv covers pieces q = (4 wv + i) * 64 + lane: pixel slot q >> 3 (row-major 12 x 20), part q & 7.
  // Pieces outside the image are never fetched (exec-masked): their LDS positions - the same for every slab of the tile - are
  // zeroed once, in both buffers; that IS the convolution's zero padding.
  unsigned poff[4];
  bool pok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = (4 * wv + i) * 64 + lane, ps = q >> 3, part = q & 7;
    const int pr = ps / RW_PW, pc = ps - pr * RW_PW;
    const int y = y0 - 2 + pr, x = x0 - 2 + pc;
    pok[i] = ps < 12 * RW_PW && y >= 0 && y < H && x >= 0 && x < W;
    poff[i] = pok[i] ? (unsigned)(((long)y * W + x) * (RW_C * 2) + part * 16) : 0u;
    if (!pok[i]) {
      const rw_u32x4 z = {0u, 0u, 0u, 0u};
      *(ROMA_LDS rw_u32x4*)(P0 + (4 * wv + i) * 1024 + lane * 16) = z;
      *(ROMA_LDS rw_u32x4*)(P0 + RW_PATCH_B + (4 * wv + i) * 1024 + lane * 16) = z;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the zeros are written before any DMA is issued
  // W slab: instruction i of wave wv covers LDS rows n = (9 wv + i) * 8 + (lane >> 3), 16-byte slot j = lane & 7 holds
  // chunk j ^ ((n >> 1) & 7) of the row's 64 k.  (n >> 1) & 7 = (36 wv + 4 i + (lane >> 4)) & 7.
  const char* const pwb_ = reinterpret_cast<const char*>(pw);
  const unsigned ldw2 = (unsigned)ldpw * 2u;
  const unsigned woff0 = (unsigned)(9 * wv * 8 + (lane >> 3)) * ldw2;
#define ROMA_RW_ISSUE_W(S)                                                                          \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 9; ++i) {                                                 \
      const unsigned c_ = (unsigned)((lane & 7) ^ ((36 * wv + 4 * i + (lane >> 4)) & 7));           \
      rw_glds16(pwb_ + (woff0 + (unsigned)(8 * i) * ldw2 + (unsigned)((S) * 128) + c_ * 16u), WB + (9 * wv + i) * 1024); \
    }                                                                                               \
  }
  // (a variable number of instructions per wave - none for a patch row outside the image: so the patch is issued BEFORE the W
  //  slab and both are awaited with vmcnt(0), never with a count)
#define ROMA_RW_ISSUE_PATCH(S, BUF)                                                                 \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
        if (pok[i]) rw_glds16(imb + (poff[i] + (unsigned)((S) * 128)), P0 + (BUF) * RW_PATCH_B + (4 * wv + i) * 1024); \
  }

  // ---------------------------------------------------------------- stencil role of this lane
  const int scol = 2 * wv + (lane >> 5);  // output column 0 .. 15
  const int sp = lane & 31;               // channel pair inside the slab
  const unsigned prd = (unsigned)(size_t)P0 + (unsigned)(scol * 128 + sp * 4);  // patch (row 0, column scol + 0), this pair
  // A-tile write of output row r: pixel px = 16 r + scol, chunk sp >> 2, bytes (sp & 3) * 4 inside the chunk.  The swizzle
  // term (px >> 1) & 7 = (scol >> 1) & 7 does not depend on r: one base register + r * 2048
  const unsigned awr0 = (unsigned)(scol * 128 + (((sp >> 2) ^ ((scol >> 1) & 7)) * 16) + (sp & 3) * 4);

  // ---------------------------------------------------------------- MFMA role of this wave: pixels [64 wm, +64) x couts [144 wn, +144)
  const int wm = wv & 1, wn = wv >> 1;
  const int l15 = lane & 15, lq = lane >> 4;
  f32x4 acc[9][4];
#pragma unroll
  for (int nb = 0; nb < 9; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // taps of the current slab: 25 x (2 channels) + bias
  f32x2 tw[25];
  f32x2 tb;
#define ROMA_RW_LOAD_TAPS(S)                                                                        \
  {                                                                                                 \
    const float* wp_ = dww + (S) * 64 + 2 * sp;                                                     \
    _Pragma("unroll") for (int t = 0; t < 25; ++t) tw[t] = *reinterpret_cast<const f32x2*>(wp_ + (long)t * RW_C); \
    tb = *reinterpret_cast<const f32x2*>(dwb + (S) * 64 + 2 * sp);                                  \
  }

  // ---------------------------------------------------------------- prologue
  ROMA_RW_ISSUE_PATCH(0, 0);
  ROMA_RW_LOAD_TAPS(0);

#pragma unroll 1
  for (int s = 0; s < RW_NS; ++s) {
    const int buf = s & 1;
    // patch(s) (issued a slab ago) and the taps of this slab have landed; every wave is done with MFMA(s - 1): the W buffer,
    // the A tile and patch buffer (s + 1) & 1 are free
    ROMA_RW_WAIT_VM(0);
    // retire the tap loads in the COMPILER's book here, before this slab's DMA is issued: it does not see the wait above, and
    // with the (conditional) DMA instructions between the loads and their first use it answers with a full vmcnt(0) in front
    // of the stencil - which would serialise the whole W slab's latency with it
#pragma unroll
    for (int t = 0; t < 25; ++t) asm volatile("" : "+v"(tw[t]));
    asm volatile("" : "+v"(tb));
    ROMA_RW_BARRIER();
    if (s + 1 < RW_NS) ROMA_RW_ISSUE_PATCH(s + 1, buf ^ 1);
    ROMA_RW_ISSUE_W(s);

    // ------------------------------------------------ stencil: 12 patch rows -> 8 output rows of (this column, this pair)
    {
      f32x2 oacc[RW_TH];  // (an output row's accumulator starts at the bias when its first patch row arrives: <= 5 live)
      const unsigned rd = prd + (unsigned)(buf * RW_PATCH_B);
#pragma unroll
      for (int R = 0; R < RW_TH + 4; ++R) {
        unsigned u0, u1, u2, u3, u4;
        asm volatile(
            "ds_read_b32 %0, %5 offset:%6\n\tds_read_b32 %1, %5 offset:%7\n\tds_read_b32 %2, %5 offset:%8\n\t"
            "ds_read_b32 %3, %5 offset:%9\n\tds_read_b32 %4, %5 offset:%10\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4)
            : "v"(rd), "n"((R * RW_PW + 0) * 128), "n"((R * RW_PW + 1) * 128), "n"((R * RW_PW + 2) * 128),
              "n"((R * RW_PW + 3) * 128), "n"((R * RW_PW + 4) * 128)
            : "memory");
        const unsigned uu[5] = {u0, u1, u2, u3, u4};
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const f32x2 v = f32x2{h16_lo(uu[dx]), h16_hi(uu[dx])};
#pragma unroll
          for (int r = 0; r < RW_TH; ++r) {
            const int dy = R - r;  // input row R = output row r + dy
            if (dy == 0 && dx == 0) oacc[r] = tb;
            if (dy >= 0 && dy < 5) oacc[r] = v * tw[dy * 5 + dx] + oacc[r];
          }
        }
        if (R >= 4) {  // output row R - 4 is complete
          const int r = R - 4;
          const unsigned pk = pack_bf16x2(fmaxf(oacc[r][0], 0.f), fmaxf(oacc[r][1], 0.f));
          *(ROMA_LDS unsigned*)(AT + awr0 + r * 2048) = pk;
        }
      }
    }
    // the W slab (and the next patch, issued before it) has landed; A tile complete
    ROMA_RW_WAIT_VM(0);
    ROMA_RW_BARRIER();
    if (s + 1 < RW_NS) ROMA_RW_LOAD_TAPS(s + 1);  // under the MFMAs

    // ------------------------------------------------ MFMA: acc[nb][mb] += W[144 wn + 16 nb ..][k] . A[64 wm + 16 mb ..][k]
    // fragment addresses: row * 128 + ((chunk ^ swz) * 16) with swz = (row >> 1) & 7 = (l15 >> 1) & 7 for every block (block
    // bases are multiples of 16 rows): one base per operand and k-step, blocks at immediate offsets of 2 KiB
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int sw16 = ((4 * kk + lq) ^ ((l15 >> 1) & 7)) * 16;
      bf16x8 bfr[4];
      {
        const lds_u8* ab = AT + (64 * wm + l15) * 128 + sw16;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) bfr[mb] = *(const ROMA_LDS bf16x8*)(ab + mb * 2048);
      }
      const unsigned wa = (unsigned)(size_t)WB + (unsigned)((144 * wn + l15) * 128 + sw16);
#pragma unroll
      for (int nb = 0; nb < 9; ++nb) {
        bf16x8 afr;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(afr) : "v"(wa), "n"(nb * 2048) : "memory");
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = mfma_h16_16x16x32(afr, bfr[mb], acc[nb][mb]);
      }
    }
  }
#undef ROMA_RW_ISSUE_W
#undef ROMA_RW_ISSUE_PATCH
#undef ROMA_RW_LOAD_TAPS

  // ---------------------------------------------------------------- epilogue: + bias, round, stage, 16-byte row stores
  ROMA_RW_WAIT_VM(0);
  lds_u8* const ST = (lds_u8*)dyn;
  bf16_t* const ob = out + (long)b * H * W * RW_C;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    ROMA_RW_BARRIER();  // MFMAs of every wave done / previous half streamed out
    if (wm == half) {
#pragma unroll
      for (int nb = 0; nb < 9; ++nb) {
        const int n0 = 144 * wn + 16 * nb + 4 * lq;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(pwb + n0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const f32x4 a = acc[nb][mb];
          rw_u32x2 q;
          q[0] = pack_bf16x2(a[0] + bv[0], a[1] + bv[1]);
          q[1] = pack_bf16x2(a[2] + bv[2], a[3] + bv[3]);
          *(ROMA_LDS rw_u32x2*)(ST + (16 * mb + l15) * RW_OPITCH + n0 * 2) = q;
        }
      }
    }
    ROMA_RW_BARRIER();
#pragma unroll
    for (int it = 0; it < 9; ++it) {
      const int q = it * 512 + tid;  // 64 px x 72 pieces of 16 B
      const int pxl = q / 72, j = q - pxl * 72;
      const int y = y0 + 4 * half + (pxl >> 4), x = x0 + (pxl & 15);
      const rw_u32x4 v = *(ROMA_LDS rw_u32x4*)(ST + pxl * RW_OPITCH + j * 16);
      if (y < H && x < W) *reinterpret_cast<rw_u32x4*>(ob + ((long)y * W + x) * RW_C + j * 8) = v;
    }
  }
}

int g_rb_wide = -1;  // roma_tuning("rb_wide", v): 1 = this kernel for C = 576 (default), 0 = dwconv5x5 + 1x1 GEMM, -1 = env ROMA_RB_WIDE

bool refiner_block_wide_supported(int Cp, int dt) { return dt == DT_BF16 && Cp == RW_C; }

// 0 = launched, 1 = not taken (the caller runs dwconv5x5 + GEMM), < 0 = error
int refiner_block_wide_try_launch(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw, long ldpw,
                                  const float* pw_b, int B, int H, int W, int Cp, int dt, hipStream_t s) {
  static const int env = getenv("ROMA_RB_WIDE") ? atoi(getenv("ROMA_RB_WIDE")) : 0;  // v0 is correct but slower than the pair: off until it wins
  if (!(g_rb_wide >= 0 ? g_rb_wide : env)) return 1;
  if (!refiner_block_wide_supported(Cp, dt) || H < 1 || W < 1 || B < 1) return 1;
  if ((long)H * W * Cp * 2 >= (1l << 32)) return 1;  // 32-bit byte offsets inside an image
  if ((reinterpret_cast<uintptr_t>(in) & 15) != 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0) return 1;
  if ((reinterpret_cast<uintptr_t>(pw) & 15) != 0 || ldpw % 8 != 0 || ldpw < Cp || ldpw > 65536) return 1;
  if ((reinterpret_cast<uintptr_t>(dw_w) & 7) != 0 || (reinterpret_cast<uintptr_t>(dw_b) & 7) != 0 ||
      (reinterpret_cast<uintptr_t>(pw_b) & 15) != 0)
    return 1;
  ROMA_REQUIRE(in != out, "refiner_block_wide: in and out must not alias");
  const int nty = (H + RW_TH - 1) / RW_TH, ntx = (W + RW_TW - 1) / RW_TW;
  const long ntiles = (long)B * nty * ntx;
  ROMA_REQUIRE(ntiles < (1l << 30), "refiner_block_wide: grid too large");
  // algorithmic work of the block: the 1x1's FLOPs (the stencil's 50 FLOP per element ride along)
  ProfScope ps("refiner_block_wide_kernel<576>", 2.0 * (double)B * H * W * (double)Cp * Cp, "flop", s);
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&refiner_block_wide_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, RW_DYN));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(refiner_block_wide_kernel, dim3((unsigned)(((ntiles + 7) / 8) * 8)), dim3(512), RW_DYN, s,
                     (const bf16_t*)in, (bf16_t*)out, dw_w, dw_b, (const bf16_t*)pw, ldpw, pw_b, B, H, W, nty, ntx, ntiles);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
