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
// v_pk_fma_f32 + ~330 other VALU per wave.  v0 (round 5, visit 2) ran the two phases back to back with all waves in lock-step:
// bit-identical to the two-kernel path but 1 372 vs 996 us at 16 x 216 x 216; v2 below de-phases two wave groups.
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
constexpr int RW_NH = RW_C / 32;              // 18 half-slabs of 32 input channels
constexpr int RW_TH = 8, RW_TW = 16;          // output tile: 8 rows x 16 columns (two wave groups x 4 rows)
constexpr int RW_PW = RW_TW + 4;              // patch: 12 rows x 20 columns of 128-byte pixels (one 64-channel slab)
constexpr int RW_PATCH_B = 32 * 1024;         // one patch buffer: 240 pixels + 16 unused slots (32 DMA instructions of 1 KiB)
constexpr int RW_WH_B = RW_C * 64;            // one W half-slab: 576 rows x 64 B = 36 KiB
constexpr int RW_TAP_B = 4096;                // one tap buffer: 26 rows (25 taps + bias) x 32 channels x f32, in 4 DMA instructions
constexpr int RW_DYN = 2 * RW_PATCH_B + 2 * RW_WH_B + 3 * RW_TAP_B;  // 151 552 B dynamic LDS
constexpr int RW_OPITCH = RW_C * 2 + 16;      // staged output pixel pitch (1168 B: 16-byte aligned, 2-way on the 8-byte writes)
static_assert(64 * RW_OPITCH <= 2 * RW_PATCH_B + RW_WH_B, "a group's staged output must not reach W buffer 1 (read until the last interval)");
static_assert(RW_DYN + 2 * 64 * 64 <= 160 * 1024, "LDS");
static_assert(64 * RW_OPITCH <= 2 * RW_PATCH_B + RW_WH_B, "staging stays below W buffer 1 and the tap ring");

__device__ __forceinline__ void rw_glds16(const char* src, lds_u8* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (ROMA_LDS void*)lds_wave_base, 16, 0, 0);
}
#define ROMA_RW_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define ROMA_RW_BARRIER()                            \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_s_barrier();                      \
  asm volatile("" ::: "memory")

// Schedule (v2, "de-phased").  The eight waves are two groups of four; waves w and w + 4 share a SIMD.  Group A owns tile rows
// 0-3, group B rows 4-7; K runs over 18 half-slabs h of 32 input channels.  Time is cut into intervals by workgroup barriers:
//     interval t = 2h     : A stencil(h)      | B MFMA(h - 1)
//     interval t = 2h + 1 : A MFMA(h)         | B stencil(h)
// so on every SIMD one wave issues the depthwise taps on the VALU while the other keeps the matrix core busy - the
// complementary overlap two streams cannot give (a GEMM workgroup owns its CU's registers; DESIGN.md "stream split").
//   * stencil(h) of a group: lane = (column 0 .. 15, channel pair 0 .. 15); 8 patch rows -> its 4 output rows, ReLU, 16-bit,
//     into the group's [64 px][32 k] A region (64-byte rows, chunk ^ ((-(px >> 2)) & 3): conflict free for the fragment reads);
//   * MFMA(h) of a group: wave = 64 pixels x 144 output channels, ONE k-step of v_mfma_f32_16x16x32: 36 MFMAs, 13 fragment reads;
//   * LDS-DMA is issued by all waves at the start of every ODD interval t = 2h + 1: W half-slab h + 1 (36 KiB into buffer
//     (h + 1) & 1, last read by B at t = 2h) and, for even h, the 64-channel patch of slab h / 2 + 1 (into the buffer B's
//     stencil left at t = 2h - 1); everything in flight is awaited with vmcnt(0) at the end of the EVEN interval t = 2h + 2 - two
//     intervals of flight time, no counting across the conditional patch pieces;
//   * a patch pixel's two 64-byte halves are swapped for odd patch columns (on the DMA's source side), so the two columns a
//     32-lane ds_read_b32 group touches fall on different bank halves.
// The tap weights (25 x 32 f32 + the bias row per half-slab) travel by LDS-DMA too, two half-slabs ahead into a ring of three
// buffers (issued by waves 4-7, which carry one W piece less), and are read into registers at the start of a stencil phase:
// they occupy registers only while they are used (as global loads prefetched under the MFMAs they cost 52 live registers
// next to the 144 accumulators and hipcc spilled 48).
// PK: the stencil's FMAs as v_pk_fma_f32 (1) or as pairs of v_fma_f32 (0) - the guide prices a packed-f32 instruction beside
// MFMAs at +22 cycles over two scalar ones; both round identically
template <int PK>
__global__ __launch_bounds__(512, 2) void refiner_block_wide_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                                    const float* __restrict__ dww, const float* __restrict__ dwb,
                                                                    const bf16_t* __restrict__ pw, long ldpw,
                                                                    const float* __restrict__ pwb, int B, int H, int W, int nty,
                                                                    int ntx, long ntiles, int dbg) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char dyn[];  // [patch 0][patch 1][W half 0][W half 1][taps x 3]: DMA targets
  __shared__ __attribute__((aligned(1024))) unsigned char atile[2 * 64 * 64];  // per group [64 px][32 k] 16-bit, swizzled chunks
  lds_u8* const P0 = (lds_u8*)dyn;
  lds_u8* const WB = (lds_u8*)dyn + 2 * RW_PATCH_B;
  lds_u8* const TB = (lds_u8*)dyn + 2 * RW_PATCH_B + 2 * RW_WH_B;

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
  const int grp = wv >> 2, wq = wv & 3;

  // ---------------------------------------------------------------- DMA descriptors (uniform base + 32-bit lane offset)
  const char* const imb = reinterpret_cast<const char*>(in + (long)b * H * W * RW_C);
  // patch: instruction i of wave wv covers pieces q = (4 wv + i) * 64 + lane: pixel slot q >> 3 (row-major 12 x 20), 16-byte
  // slot q & 7 of the pixel, which holds part (q & 7) ^ 4 for odd patch columns.  Pieces outside the image are never fetched
  // (exec-masked); their positions - the same for every slab - are zeroed once in both buffers: the zero padding.
  unsigned poff[4];
  bool pok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = (4 * wv + i) * 64 + lane, ps = q >> 3;
    const int pr = ps / RW_PW, pc = ps - pr * RW_PW;
    const int part = (q & 7) ^ ((pc & 1) << 2);
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
  // W half-slab: instruction k covers LDS rows 16 k + (lane >> 2) (64-byte rows), slot lane & 3 holds chunk
  // (lane & 3) ^ ((-(row >> 2)) & 3) = (lane & 3) ^ ((-(lane >> 4)) & 3); wave wv issues k = wv, wv + 8, wv + 16, wv + 24 and,
  // waves 0-3, wv + 32
  const char* const pwb_ = reinterpret_cast<const char*>(pw);
  const unsigned ldw2 = (unsigned)ldpw * 2u;
  const unsigned woff0 = (unsigned)(16 * wv + (lane >> 2)) * ldw2 + (unsigned)(((lane & 3) ^ ((0 - (lane >> 4)) & 3)) * 16);
#define ROMA_RW_ISSUE_W(HH)                                                                                         \
  {                                                                                                                 \
    lds_u8* const wdst_ = WB + ((HH) & 1) * RW_WH_B + wv * 1024;                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                   \
        rw_glds16(pwb_ + (woff0 + (unsigned)(128 * i) * ldw2 + (unsigned)((HH) * 64)), wdst_ + i * 8192);           \
    if (wv < 4) rw_glds16(pwb_ + (woff0 + 512u * ldw2 + (unsigned)((HH) * 64)), wdst_ + 4 * 8192);                  \
  }
#define ROMA_RW_ISSUE_PATCH(S)                                                                                      \
  {                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                   \
        if (pok[i]) rw_glds16(imb + (poff[i] + (unsigned)((S) * 128)), P0 + ((S) & 1) * RW_PATCH_B + (4 * wv + i) * 1024); \
  }

  // ---------------------------------------------------------------- stencil role: column scol, channel pair sp of the half-slab
  const int scol = 4 * wq + (lane >> 4);
  const int sp = lane & 15;
  // patch address of (patch row 4 grp + R, patch column scol + dx), half e = sub ^ ((scol + dx) & 1) of the pixel
  const unsigned prd0 = (unsigned)(size_t)P0 + (unsigned)((4 * grp * RW_PW + scol) * 128 + sp * 4);
  // A-region write of output row r: px = 16 r + scol; (-(px >> 2)) & 3 = (-(scol >> 2)) & 3 for every r
  lds_u8* const ATg = (lds_u8*)atile + grp * 4096;
  const unsigned awr0 = (unsigned)(scol * 64 + (((sp >> 2) ^ ((0 - (scol >> 2)) & 3)) * 16) + (sp & 3) * 4);

  // ---------------------------------------------------------------- MFMA role: pixels of the group x couts [144 wq, +144)
  const int l15 = lane & 15, lq = lane >> 4;
  const int fsw16 = ((lq ^ ((0 - (l15 >> 2)) & 3)) * 16);  // chunk slot of this lane's k block in a 64-byte row
  f32x4 acc[9][4];
#pragma unroll
  for (int nb = 0; nb < 9; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // taps of half-slab HH -> tap buffer HH % 3: wave 4 + j fetches rows 8 j .. 8 j + 7 ([row][32 ch] f32 = 128 B; row 25 = bias)
  const int tap_row = 8 * wq + (lane >> 3);
  const char* const tap_src = tap_row < 25 ? reinterpret_cast<const char*>(dww) + ((long)tap_row * RW_C * 4 + (lane & 7) * 16)
                                           : reinterpret_cast<const char*>(dwb) + (lane & 7) * 16;
#define ROMA_RW_ISSUE_TAPS(HH)                                                                                          \
  if (grp == 1 && tap_row < 26) rw_glds16(tap_src + (HH) * 128, TB + ((HH) % 3) * RW_TAP_B + wq * 1024);
  f32x2 tw[25];
  f32x2 tb;
  const unsigned tap_rd = (unsigned)(size_t)TB + (unsigned)(sp * 8);
#define ROMA_RW_T4(I0)                                                                                                  \
  asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%6\n\tds_read_b64 %2, %4 offset:%7\n\t"          \
               "ds_read_b64 %3, %4 offset:%8"                                                                            \
               : "=&v"(tw[I0]), "=&v"(tw[I0 + 1]), "=&v"(tw[I0 + 2]), "=&v"(tw[I0 + 3])                                 \
               : "v"(trd_), "n"((I0) * 128), "n"((I0 + 1) * 128), "n"((I0 + 2) * 128), "n"((I0 + 3) * 128)              \
               : "memory")
#define ROMA_RW_READ_TAPS(HH)                                                                                           \
  {                                                                                                                     \
    const unsigned trd_ = tap_rd + (unsigned)(((HH) % 3) * RW_TAP_B);                                                   \
    ROMA_RW_T4(0); ROMA_RW_T4(4); ROMA_RW_T4(8); ROMA_RW_T4(12); ROMA_RW_T4(16); ROMA_RW_T4(20);                        \
    asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"                 \
                 : "=&v"(tw[24]), "=&v"(tb) : "v"(trd_), "n"(24 * 128), "n"(25 * 128) : "memory");                      \
    _Pragma("unroll") for (int t = 0; t < 25; ++t) asm volatile("" : "+v"(tw[t]));                                      \
  }

  // stencil(h): 8 patch rows -> the group's 4 output rows of (column scol, pair sp); the reads of row R + 1 are in flight
  // under the FMAs of row R
#define ROMA_RW_RD5(U, RR)                                                                                              \
  asm volatile("ds_read_b32 %0, %5 offset:%7\n\tds_read_b32 %1, %6 offset:%7\n\tds_read_b32 %2, %5 offset:%8\n\t"        \
               "ds_read_b32 %3, %6 offset:%8\n\tds_read_b32 %4, %5 offset:%9"                                             \
               : "=&v"(U[0]), "=&v"(U[1]), "=&v"(U[2]), "=&v"(U[3]), "=&v"(U[4])                                          \
               : "v"(rde), "v"(rdo), "n"((RR) * RW_PW * 128), "n"((RR) * RW_PW * 128 + 256), "n"((RR) * RW_PW * 128 + 512) \
               : "memory")
#define ROMA_RW_STENCIL(HH)                                                                                             \
  {                                                                                                                     \
    const int sub_ = (HH) & 1, pb_ = ((HH) >> 1) & 1;                                                                   \
    const int e_ = sub_ ^ (scol & 1);                                                                                   \
    const unsigned rde = prd0 + (unsigned)(pb_ * RW_PATCH_B + e_ * 64);            /* dx = 0, 2, 4 */                   \
    const unsigned rdo = prd0 + (unsigned)(pb_ * RW_PATCH_B + 128 + (1 - e_) * 64); /* dx = 1, 3 */                     \
    f32x2 oacc[4];                                                                                                      \
    unsigned ua[5], ub[5];                                                                                              \
    if (!(dbg & 8)) ROMA_RW_READ_TAPS(HH);                                                                              \
    ROMA_RW_RD5(ua, 0);                                                                                                 \
    _Pragma("unroll") for (int R = 0; R < 8; ++R) {                                                                     \
      unsigned(&cur)[5] = (R & 1) ? ub : ua;                                                                            \
      unsigned(&nxt)[5] = (R & 1) ? ua : ub;                                                                            \
      if (R < 7) {                                                                                                      \
        switch (R) {                                                                                                    \
          case 0: ROMA_RW_RD5(nxt, 1); break;                                                                           \
          case 1: ROMA_RW_RD5(nxt, 2); break;                                                                           \
          case 2: ROMA_RW_RD5(nxt, 3); break;                                                                           \
          case 3: ROMA_RW_RD5(nxt, 4); break;                                                                           \
          case 4: ROMA_RW_RD5(nxt, 5); break;                                                                           \
          case 5: ROMA_RW_RD5(nxt, 6); break;                                                                           \
          default: ROMA_RW_RD5(nxt, 7); break;                                                                          \
        }                                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4])::"memory"); \
      } else {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4])::"memory"); \
      }                                                                                                                 \
      /* read order was dx = 0, 1, 2, 3, 4 -> cur[0 .. 4] */                                                            \
      _Pragma("unroll") for (int dx = 0; dx < 5; ++dx) {                                                                \
        const f32x2 v = f32x2{h16_lo(cur[dx]), h16_hi(cur[dx])};                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                 \
          const int dy = R - r;                                                                                         \
          if (dy == 0 && dx == 0) oacc[r] = tb;                                                                         \
          if (dy >= 0 && dy < 5) {                                                                                      \
            if constexpr (PK != 0) {                                                                                    \
              oacc[r] = v * tw[dy * 5 + dx] + oacc[r];                                                                  \
            } else {                                                                                                    \
              oacc[r][0] = __builtin_fmaf(v[0], tw[dy * 5 + dx][0], oacc[r][0]);                                        \
              oacc[r][1] = __builtin_fmaf(v[1], tw[dy * 5 + dx][1], oacc[r][1]);                                        \
            }                                                                                                           \
          }                                                                                                             \
        }                                                                                                               \
      }                                                                                                                 \
      if (R >= 4) {                                                                                                     \
        const int r = R - 4;                                                                                            \
        const unsigned pk = pack_bf16x2(fmaxf(oacc[r][0], 0.f), fmaxf(oacc[r][1], 0.f));                                \
        *(ROMA_LDS unsigned*)(ATg + awr0 + r * 1024) = pk;                                                              \
      }                                                                                                                 \
    }                                                                                                                   \
  }

  // MFMA(h): acc[nb][mb] += W[144 wq + 16 nb ..][32 k] . A[16 mb ..][32 k]; W fragments (asm reads from the DMA target) three ahead
#define ROMA_RW_RDW(DST, NB) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(DST) : "v"(wa_), "n"((NB) * 1024) : "memory")
#define ROMA_RW_MF4(AF, NB)                                                                                             \
  _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) acc[NB][mb] = mfma_h16_16x16x32(AF, bfr_[mb], acc[NB][mb]);
#define ROMA_RW_MFMA(HH)                                                                                                \
  {                                                                                                                     \
    const unsigned wa_ = (unsigned)(size_t)WB + (unsigned)(((HH) & 1) * RW_WH_B + (144 * wq + l15) * 64 + fsw16);       \
    bf16x8 bfr_[4];                                                                                                     \
    {                                                                                                                   \
      const lds_u8* ab_ = ATg + l15 * 64 + fsw16;                                                                       \
      _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) bfr_[mb] = *(const ROMA_LDS bf16x8*)(ab_ + mb * 1024);           \
    }                                                                                                                   \
    bf16x8 w0_, w1_, w2_;                                                                                               \
    ROMA_RW_RDW(w0_, 0); ROMA_RW_RDW(w1_, 1); ROMA_RW_RDW(w2_, 2);                                                      \
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w0_)::"memory"); ROMA_RW_MF4(w0_, 0); ROMA_RW_RDW(w0_, 3);               \
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w1_)::"memory"); ROMA_RW_MF4(w1_, 1); ROMA_RW_RDW(w1_, 4);               \
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w2_)::"memory"); ROMA_RW_MF4(w2_, 2); ROMA_RW_RDW(w2_, 5);               \
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w0_)::"memory"); ROMA_RW_MF4(w0_, 3); ROMA_RW_RDW(w0_, 6);               \
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w1_)::"memory"); ROMA_RW_MF4(w1_, 4); ROMA_RW_RDW(w1_, 7);               \
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w2_)::"memory"); ROMA_RW_MF4(w2_, 5); ROMA_RW_RDW(w2_, 8);               \
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w0_)::"memory"); ROMA_RW_MF4(w0_, 6);                                    \
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w1_)::"memory"); ROMA_RW_MF4(w1_, 7);                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w2_)::"memory"); ROMA_RW_MF4(w2_, 8);                                    \
  }

  // epilogue pieces: a group's accumulators + bias -> 16-bit -> staging [64 px][1168 B]; then every wave streams rows out
  lds_u8* const ST = (lds_u8*)dyn;
  bf16_t* const ob = out + (long)b * H * W * RW_C;
#define ROMA_RW_STAGE()                                                                                                 \
  {                                                                                                                     \
    _Pragma("unroll") for (int nb = 0; nb < 9; ++nb) {                                                                  \
      const int n0 = 144 * wq + 16 * nb + 4 * lq;                                                                       \
      const f32x4 bv = *reinterpret_cast<const f32x4*>(pwb + n0);                                                       \
      _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) {                                                                \
        const f32x4 a = acc[nb][mb];                                                                                    \
        rw_u32x2 q;                                                                                                     \
        q[0] = pack_bf16x2(a[0] + bv[0], a[1] + bv[1]);                                                                 \
        q[1] = pack_bf16x2(a[2] + bv[2], a[3] + bv[3]);                                                                 \
        *(ROMA_LDS rw_u32x2*)(ST + (16 * mb + l15) * RW_OPITCH + n0 * 2) = q;                                           \
      }                                                                                                                 \
    }                                                                                                                   \
  }
#define ROMA_RW_COPYOUT(HALF)                                                                                           \
  {                                                                                                                     \
    _Pragma("unroll") for (int it = 0; it < 9; ++it) {                                                                  \
      const int q = it * 512 + tid; /* 64 px x 72 pieces of 16 B */                                                     \
      const int pxl = q / 72, j = q - pxl * 72;                                                                         \
      const int y = y0 + 4 * (HALF) + (pxl >> 4), x = x0 + (pxl & 15);                                                  \
      const rw_u32x4 v = *(ROMA_LDS rw_u32x4*)(ST + pxl * RW_OPITCH + j * 16);                                          \
      if (y < H && x < W) *reinterpret_cast<rw_u32x4*>(ob + ((long)y * W + x) * RW_C + j * 8) = v;                      \
    }                                                                                                                   \
  }

  // ---------------------------------------------------------------- prologue: patch 0, W half 0, taps of halves 0 and 1
  ROMA_RW_ISSUE_PATCH(0);
  ROMA_RW_ISSUE_W(0);
  ROMA_RW_ISSUE_TAPS(0);
  ROMA_RW_ISSUE_TAPS(1);
  ROMA_RW_WAIT_VM(0);
  ROMA_RW_BARRIER();

  if (grp == 0) {
    // ================================================================ group A: stencil at even intervals, MFMA at odd ones
    if (!(dbg & 1)) ROMA_RW_STENCIL(0);                  // t = 0
    ROMA_RW_BARRIER();
#pragma unroll 1
    for (int h = 0; h < RW_NH - 1; ++h) {
      // t = 2h + 1: DMA for half-slab h + 1, MFMA(h)
      if (!(dbg & 4)) {
        if ((h & 1) == 0 && h + 2 < RW_NH) ROMA_RW_ISSUE_PATCH((h >> 1) + 1);
        ROMA_RW_ISSUE_W(h + 1);
      }
      if (!(dbg & 2)) ROMA_RW_MFMA(h);
      ROMA_RW_BARRIER();
      // t = 2h + 2: stencil(h + 1); then everything issued at t = 2h + 1 has landed
      if (!(dbg & 1)) ROMA_RW_STENCIL(h + 1);
      ROMA_RW_WAIT_VM(0);
      ROMA_RW_BARRIER();
    }
    if (!(dbg & 2)) ROMA_RW_MFMA(RW_NH - 1);             // t = 35
    ROMA_RW_BARRIER();
    if (!(dbg & 16)) ROMA_RW_STAGE();                     // t = 36: A's rows into the staging area (B is in its last MFMA: W buffer 1, its A region)
    ROMA_RW_BARRIER();
    if (!(dbg & 16)) ROMA_RW_COPYOUT(0);                     // t = 37
    ROMA_RW_BARRIER();
    ROMA_RW_BARRIER();                      // t = 38: B stages
    if (!(dbg & 16)) ROMA_RW_COPYOUT(1);                     // t = 39
  } else {
    // ================================================================ group B: one interval behind
    ROMA_RW_BARRIER();                      // t = 0: nothing to do yet
#pragma unroll 1
    for (int h = 0; h < RW_NH; ++h) {
      // t = 2h + 1: DMA for half-slab h + 1 (and the taps of h + 2), stencil(h)
      if (h + 1 < RW_NH && !(dbg & 4)) {
        if ((h & 1) == 0 && h + 2 < RW_NH) ROMA_RW_ISSUE_PATCH((h >> 1) + 1);
        ROMA_RW_ISSUE_W(h + 1);
        if (h + 2 < RW_NH) ROMA_RW_ISSUE_TAPS(h + 2);
      }
      if (!(dbg & 1)) ROMA_RW_STENCIL(h);
      ROMA_RW_BARRIER();
      // t = 2h + 2: MFMA(h); then everything in flight has landed
      if (!(dbg & 2)) ROMA_RW_MFMA(h);
      ROMA_RW_WAIT_VM(0);
      ROMA_RW_BARRIER();
    }
    if (!(dbg & 16)) ROMA_RW_COPYOUT(0);                     // t = 37: A's rows
    ROMA_RW_BARRIER();
    if (!(dbg & 16)) ROMA_RW_STAGE();                     // t = 38
    ROMA_RW_BARRIER();
    if (!(dbg & 16)) ROMA_RW_COPYOUT(1);                     // t = 39
  }
}

int g_rb_wide = -1;  // roma_tuning("rb_wide", v): 1 = this kernel for C = 576, 0 = dwconv5x5 + 1x1 GEMM (default: the fused kernel measured slower), 2 / 3 = this kernel with the scalar / packed stencil, -1 = env ROMA_RB_WIDE

bool refiner_block_wide_supported(int Cp, int dt) { return dt == DT_BF16 && Cp == RW_C; }

// 0 = launched, 1 = not taken (the caller runs dwconv5x5 + GEMM), < 0 = error
int refiner_block_wide_try_launch(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw, long ldpw,
                                  const float* pw_b, int B, int H, int W, int Cp, int dt, hipStream_t s, bool force) {
  static const int env = getenv("ROMA_RB_WIDE") ? atoi(getenv("ROMA_RB_WIDE")) : 0;  // measured slower than dwconv5x5 + ws1x1 (1 306 vs 978 us): off
  if (!force && !(g_rb_wide >= 0 ? g_rb_wide : env)) return 1;
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
#ifdef ROMA_TOOLS_BUILD  // ablations (tools/bench_refiner_wide.py): 1 no stencil, 2 no MFMA, 4 no DMA after the prologue, 8 no tap reads, 16 no epilogue
  static const int dbg = getenv("ROMA_RBW_DBG") ? atoi(getenv("ROMA_RBW_DBG")) : 0;
#else  // the shipped libraries never take ablation bits from the environment (they produce wrong outputs by design)
  constexpr int dbg = 0;
#endif
  static const int pk_env = getenv("ROMA_RB_WIDE_PK") ? atoi(getenv("ROMA_RB_WIDE_PK")) : 1;
  const int mode = g_rb_wide >= 2 ? g_rb_wide : 0;  // roma_tuning("rb_wide", 2 / 3): force the scalar / packed stencil (A/B)
  const bool pk = mode == 3 ? true : (mode == 2 ? false : pk_env != 0);
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&refiner_block_wide_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, RW_DYN));
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&refiner_block_wide_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, RW_DYN));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const dim3 grid((unsigned)(((ntiles + 7) / 8) * 8));
  if (pk)
    hipLaunchKernelGGL(refiner_block_wide_kernel<1>, grid, dim3(512), RW_DYN, s, (const bf16_t*)in, (bf16_t*)out, dw_w, dw_b,
                       (const bf16_t*)pw, ldpw, pw_b, B, H, W, nty, ntx, ntiles, dbg);
  else
    hipLaunchKernelGGL(refiner_block_wide_kernel<0>, grid, dim3(512), RW_DYN, s, (const bf16_t*)in, (bf16_t*)out, dw_w, dw_b,
                       (const bf16_t*)pw, ldpw, pw_b, B, H, W, nty, ntx, ntiles, dbg);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
