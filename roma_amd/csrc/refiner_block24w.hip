// Fused depthwise-5x5 (+folded BN, ReLU) -> 1x1 convolution for the stride-1 ConvRefiner (C = 24) - the "wave-private"
// form of refiner_block_kernel<24> (refiner_block.hip).  romatch/models/matcher.py:106-122.
//
// refiner_block_kernel<24> shares everything across its four waves - the input ring, the depthwise-output tile Xt, the
// output tile Ot - and pays two workgroup barriers per image row for it; its waves issue 45 % of their cycles and wait 36 %
// (profiles/r03_pmc_sq_summary.json).  With 24 channels a single wave can own ALL channels of its pixels, so nothing has to
// be shared (the recipe of dwconv5x5_ring_kernel, which issues 55 % of its cycles with no barrier at all):
//
//   * a wave = 40 output columns x all 24 channels x a strip of rows; lane = (4 channels, 4 columns): 6 channel groups x
//     10 column quads = 60 lanes (the workgroup kernel: 216 of 256), through the lane table below (LDS bank conflicts);
//   * per input row the wave needs 44 pixels x 48 B: three `global_load_lds_dwordx4` into its own NR = 4 row ring.  After
//     every 4 pixels (12 pieces) one 16-byte piece of the row stays empty, so that the column
//     quads of a `ds_read_b64` half-wave start 52 dwords apart (0, 52, 40, 28, 16 mod 64 - five disjoint 12-dword runs;
//     the natural 48-dword pitch puts quad 4 on quad 0's banks);
//   * the depthwise output row goes to a wave-private Xt[40 pixels][32 k] (bf16, 80-byte rows as in the workgroup kernel),
//     two 32-pixel MFMA blocks (the second one 8 pixels + zeros) x 2 k-steps against the 1x1 weights held in registers,
//     bias in the accumulator init; v_permlane32_swap pairs the half-waves and the row leaves in three 16-byte stores per
//     lane straight from the accumulators (until round 5: through a wave-private LDS tile Ot[40][24], 6 ds_write_b64 +
//     2 ds_read_b128 + a wait per row);
//   * the only thing that orders anything is the wave's own counted `s_waitcnt vmcnt` (allowance = the younger DMA) and the
//     in-order LDS pipeline: NO barrier after the weights have been staged, the waves drift freely, and a wave whose tile is
//     off the image simply leaves.
//   * arithmetic and its order are those of refiner_block_kernel<24>: results are bit-identical (tests).
#include <stdlib.h>

#include <algorithm>

#include "gemm.h"  // DT_*
#include "refiner_block.h"

namespace roma {

typedef __attribute__((ext_vector_type(2))) float f32x2;
#define ROMA_LDS __attribute__((address_space(3)))
typedef ROMA_LDS unsigned char lds_u8;
typedef ROMA_LDS float lds_f32;
typedef ROMA_LDS f32x4 lds_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef ROMA_LDS u32x4_t lds_u32x4;
typedef ROMA_LDS u32x2_t lds_u32x2;

__device__ __attribute__((aligned(256))) unsigned int g_rbw_zero_page[64];  // source of every out-of-image / gap piece

constexpr int RBW_C = 24;
constexpr int RBW_NR = 4;         // ring rows per wave (the DMA runs three rows ahead)
constexpr int RBW_ROWB = 3072;    // bytes per ring row: 192 pieces of 16 B (44 pixels x 3 + 10 gaps = 142 used)
constexpr int RBW_PXW = 40;       // output columns per wave
constexpr int RBW_XROW = 80;      // bytes per Xt pixel row: 32 k x 2 B + 16 (conflict-free 16-byte MFMA fragment reads)
constexpr int RBW_XT = 64 * RBW_XROW + 32;   // two 32-pixel MFMA blocks; rows 40 .. 63 stay zero; + the second block's shift
constexpr int RBW_RING = 4 * RBW_NR * RBW_ROWB;           // 48 KiB per workgroup
constexpr int RBW_WORK = 4 * RBW_XT;                      // 20 KiB

// LDS banks (round 5; MI355X_MICROARCH.md, LDS table).  Three streams depend on which lane computes which (column quad xq,
// channel group cg) - the 8 ring reads of a row (ds_read_b64: half-waves, 64 banks), the 4 Xt writes (ds_write_b64: groups of
// 16 consecutive lanes, 32 banks = 16 units of 8 B) and, through the Xt layout, the MFMA operand reads (ds_read_b128: lane
// groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}, 16 units of 16 B).  lane = 6 xq + cg had a 2-way conflict on every ring
// read (quad 5's groups 0, 1 sit in the first half-wave, on quad 0's banks), on three of the four groups of every Xt write and
// on every Ot write: share 0.152 (profiles/r04_pmc_sq_summary.json).  Now:
//   * ring: the quads whose 12-dword runs are disjoint are {0 .. 4} and {5 .. 9} - one set per half-wave;
//   * Xt: the write unit of (xq, cg) is 8 [xq odd] + cg mod 16 with the plain 80-byte rows - two windows of six, twelve lanes
//     of sixteen at best.  A row's 16 pad bytes allow a 16-byte shift: rows of the quads with an odd number of bits in
//     (xq & 7) - {1, 2, 4, 7}, one of the two lane groups of the operand read, which therefore stays conflict free - start
//     16 bytes later, and the second MFMA block (quads 8, 9; read by its own instruction) 32 bytes later.  Windows: quads
//     0 6: 0-5, 2 4: 2-7, 3 5: 8-13, 1 7: 10-15, 8: 4-9, 9: 14-3.  The table fills every 16-lane group with distinct units,
//     except lanes 16-31 (quads 2 and 4 share a window and, with quad 0, supply three lanes per unit 2-5 to the two groups of
//     their half-wave): one extra cycle per Xt write, the minimum.
// Entry = 8 xq + cg; 255 = idle lane.  tests/test_cpu_oracle.py checks cover and conflict count of this table.
__device__ const unsigned char g_rbw_lane_map[64] = {
    0,  1,  2,  3,  4,  5,  24, 25, 26, 27, 28, 29, 20, 21, 12, 13,   // quad 0 | quad 3 | quad 2: 4 5 | quad 1: 4 5
    8,  9,  10, 11, 16, 17, 18, 19, 32, 33, 34, 35, 36, 37, 255, 255, // quad 1: 0-3 | quad 2: 0-3 | quad 4
    48, 49, 50, 51, 52, 53, 56, 57, 58, 59, 60, 61, 66, 67, 68, 69,   // quad 6 | quad 7 | quad 8: 2-5
    40, 41, 42, 43, 44, 45, 64, 65, 72, 73, 74, 75, 76, 77, 255, 255  // quad 5 | quad 8: 0 1 | quad 9
};
// byte offset of Xt pixel row p (0 .. 63)
__device__ __forceinline__ int rbw_xt_row(int p) { return p * RBW_XROW + 16 * ((0x96 >> ((p >> 2) & 7)) & 1) + (p >= 32 ? 32 : 0); }
constexpr int RBW_WSM = 26 * RBW_C * 4;                   // depthwise taps + bias, f32
static_assert(RBW_RING + RBW_WORK + RBW_WSM <= 80 * 1024, "two workgroups per CU");
static_assert(3 * (RBW_NR - 1) <= 63, "vmcnt is a 6-bit counter");

#define ROMA_RBW_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

__device__ __forceinline__ void rbw_glds16(const char* src, lds_u8* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (ROMA_LDS void*)lds_wave_base, 16, 0, 0);
}

// FINAL (round 5): the LAST block of a ConvRefiner.  Its 1x1 convolution and out_conv are one composed C -> 3 map (model.hip,
// RefinerW::oc_w); `pw` then holds 8 rows - rows 0-2 the 16-bit head of the composed weights, rows 4-6 their 16-bit remainder
// (hi + lo carries ~16 significant bits through the MFMA, which has 29 idle rows anyway) - `pwb` the composed bias in [0, 3),
// and instead of a block output the kernel writes delta[pixel] = {d flow x, d flow y, d certainty, 0} (f32x4): no Ot, no
// C-channel row stores, and refiner_out's pass over the block output disappears (refiner_apply_delta_kernel adds the deltas).
template <bool FINAL>
__global__ __launch_bounds__(256, 2) void refiner_block24_wave_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                                      const float* __restrict__ dww, const float* __restrict__ dwb,
                                                                      const bf16_t* __restrict__ pw, long ldpw,
                                                                      const float* __restrict__ pwb, int B, int H, int W, int SY,
                                                                      int nxg, long ntasks, f32x4* __restrict__ delta) {
  constexpr int NR = RBW_NR, CP = RBW_C;
  // Distinct LDS objects on purpose (refiner_block.hip): `ring` is the DMA target and is read with inline asm only; the
  // others are ordinary code, which the compiler then does not order behind the in-flight DMA.  (Ot as a slice of the Xt
  // object got an `s_waitcnt vmcnt(0)` in front of its first write of every row - the whole DMA queue drained; as an
  // object of its own it does not.  tests/test_cpu_oracle.py audits the loop for it.)
  __shared__ __attribute__((aligned(1024))) unsigned char ring[RBW_RING];
  __shared__ __attribute__((aligned(16))) unsigned char xtb[4 * RBW_XT];
  __shared__ __attribute__((aligned(16))) float wsmb[26 * RBW_C];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  lds_f32* const wsm = (lds_f32*)wsmb;  // [26][24]
  lds_u8* const Xt = (lds_u8*)xtb + wv * RBW_XT;

  // ---- one-time staging: the tap table (shared, read-only afterwards) and this wave's zeroed Xt
  for (int i = tid; i < 26 * (CP / 4); i += 256)
    *(lds_f32x4*)(wsm + i * 4) =
        *reinterpret_cast<const f32x4*>(i < 25 * (CP / 4) ? dww + (long)i * 4 : dwb + (long)(i - 25 * (CP / 4)) * 4);
  for (int i = lane; i < RBW_XT / 16; i += 64) *(lds_u32x4*)(Xt + i * 16) = u32x4_t{0u, 0u, 0u, 0u};
  __syncthreads();  // the only barrier of the kernel

  // ---- this wave's task: (image, strip, 40-column tile), tiles fastest (neighbouring waves read neighbouring columns);
  // each XCD owns a contiguous band of workgroups (the row halos hit its own L2)
  const long nwg = (ntasks + 3) / 4, wg_per_xcd = (nwg + 7) / 8;
  const long lw = (long)(blockIdx.x % 8) * wg_per_xcd + blockIdx.x / 8;
  const long task = lw * 4 + wv;
  if (lw >= nwg || task >= ntasks) return;  // (no barriers below: a wave may leave on its own)
  const int xg = (int)(task % nxg);
  long r = task / nxg;
  const int yt = (H + SY - 1) / SY;
  const int ys = (int)(r % yt) * SY;
  const int b = (int)(r / yt);
  const int sy = min(SY, H - ys);
  const int T = sy + 4;  // input rows ys - 2 .. ys + sy + 1

  unsigned lm = g_rbw_lane_map[lane];
  // retire the load in the compiler's book HERE (its wait-count pass does not see the inline-asm waits below: a load whose
  // first consumer it places inside the row loop would get an s_waitcnt vmcnt(0) there - the whole DMA queue drained per row;
  // tests/test_cpu_oracle.py audits the loop for it)
  asm volatile("" : "+v"(lm));
  const bool active = lm != 255u;
  const int cg = active ? (int)(lm & 7u) : 0, xq = active ? (int)(lm >> 3) : 0;
  const int c = cg * 4;
  const int xw0 = xg * RBW_PXW;            // first output column of the wave
  const int x0 = xw0 - 2;                  // image column of ring pixel 0
  const int npw = min(RBW_PXW, W - xw0);   // valid output columns of the wave (>= 1)

  // ---- 1x1 weights of all 24 output channels: A operand (row = channel, 8 consecutive k per lane), rows / k >= 24 zero
  u32x4_t wA[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int k0 = ks * 16 + hh * 8;
    wA[ks] = u32x4_t{0u, 0u, 0u, 0u};
    if (l31 < (FINAL ? 8 : CP) && k0 < CP) wA[ks] = *reinterpret_cast<const u32x4_t*>(pw + (long)l31 * ldpw + k0);
  }
  f32x4 pbias[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) pbias[g] = *reinterpret_cast<const f32x4*>(pwb + 8 * g + 4 * hh);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(wA[ks]));
#pragma unroll
  for (int g = 0; g < 3; ++g) asm volatile("" : "+v"(pbias[g]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this wave in flight before the counted DMA stream starts

  // ---- DMA descriptors: piece k = 64 i + lane of a row; 13 pieces per 4 pixels (12 data + 1 gap)
  const char* zsrc = reinterpret_cast<const char*>(g_rbw_zero_page);
  const char* inb = reinterpret_cast<const char*>(in + ((long)b * H * W) * CP);
  // Round 6 (dwconv_ring.hip): the DMA goes through a buffer descriptor over image b - a 32-bit lane offset fixed for the strip
  // (0x80000000 for gap pieces and columns outside the image: beyond num_records, the hardware returns zeros) plus the row
  // offset in an SGPR; rows above / below the image take the marker for every lane behind a wave-uniform branch.  No VALU in
  // the issue, no per-row selects, no 64-bit lane pointers.
  const long pitch = (long)W * CP * 2;  // (H * pitch < 2^31: checked by the launcher)
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(inb), 0, (int)((long)H * pitch), 0x00020000);
  unsigned voff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int k = 64 * i + lane, g13 = k / 13, r13 = k - g13 * 13;
    const int p = 4 * g13 + r13 / 3, part = r13 % 3;
    const int x = x0 + p;
    const bool ok = r13 != 12 && p < RBW_PXW + 4 && x >= 0 && x < W;
    voff[i] = ok ? (unsigned)(x * CP * 2 + part * 16) : 0x80000000u;
  }
  unsigned vout = 0x80000000u;  // every lane outside: rows above / below the image
  asm volatile("" : "+v"(vout));
  lds_u8* const myring = (lds_u8*)ring + wv * (NR * RBW_ROWB);
#define ROMA_RBW_BL16(VOFF, SOFF, DST) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (ROMA_LDS void*)(DST), 16, (int)(VOFF), (int)(SOFF), 0, 0)
#define ROMA_RBW_ISSUE(RROW, SLOT)                                                                  \
  {                                                                                                 \
    const int yy_ = ys - 2 + (RROW);                                                                \
    if ((RROW) < T && yy_ >= 0 && yy_ < H) { /* wave-uniform */                                     \
      const int so_ = yy_ * (int)pitch;                                                             \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) ROMA_RBW_BL16(voff[i], so_, myring + (SLOT) * RBW_ROWB + i * 1024); \
    } else {                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) ROMA_RBW_BL16(vout, 0, myring + (SLOT) * RBW_ROWB + i * 1024); \
    }                                                                                               \
  }

  const f32x4 bx = *(lds_f32x4*)(wsm + 25 * CP + c);
  const f32x2 bias0 = f32x2{bx[0], bx[1]}, bias1 = f32x2{bx[2], bx[3]};
  f32x2 acc[5][4][2];
#pragma unroll
  for (int s5 = 0; s5 < 5; ++s5)
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      acc[s5][px][0] = bias0;
      acc[s5][px][1] = bias1;
    }
  const unsigned rd0 = (unsigned)(size_t)myring + (unsigned)(xq * 208 + cg * 8);
  bf16_t* const obase = out + ((long)b * H * W) * CP;
  int xtw0 = rbw_xt_row(xq * 4) + cg * 8;                               // this lane's Xt write position (pixel 4 xq)
  int xtr0 = rbw_xt_row(l31) + hh * 16, xtr1 = rbw_xt_row(32 + l31) + hh * 16;  // its operand rows of the two MFMA blocks
  // opaque: hipcc otherwise folds the 0 / 16 / 32-byte shifts into SELECTS OF POINTERS, the LDS lowering then no longer knows
  // which LDS object an access belongs to, and every Xt access of the row loop gets an s_waitcnt vmcnt(0) in front of it
  // ("may alias the DMA target") - the whole DMA queue drained twice per row (the audit in tests/test_cpu_oracle.py)
  asm volatile("" : "+v"(xtw0), "+v"(xtr0), "+v"(xtr1));

#pragma unroll
  for (int rr = 0; rr < NR - 1; ++rr) ROMA_RBW_ISSUE(rr, rr);

  // Round 6: the ring reads of input row t + 1 are issued behind row t's MFMAs (the converted row and the MFMA operands are
  // dead by then - issued behind the FMAs, next to the 32 output accumulators, they spilled) and waited for at the top of the
  // next iteration: the packing and the stores of row t cover the LDS round trip that used to open every row.  `cr` is written asynchronously: nothing may touch it between the read and
  // the wait (tools/audit_asm_reads.py).
  unsigned long long cr[8];
#define ROMA_RBW_READ(RA)                                                                                              \
  asm volatile(                                                                                                        \
      "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:48\n\tds_read_b64 %2, %8 offset:96\n\t"                   \
      "ds_read_b64 %3, %8 offset:144\n\tds_read_b64 %4, %8 offset:208\n\tds_read_b64 %5, %8 offset:256\n\t"      \
      "ds_read_b64 %6, %8 offset:304\n\tds_read_b64 %7, %8 offset:352"                                               \
      : "=&v"(cr[0]), "=&v"(cr[1]), "=&v"(cr[2]), "=&v"(cr[3]), "=&v"(cr[4]), "=&v"(cr[5]), "=&v"(cr[6]), "=&v"(cr[7]) \
      : "v"(RA)                                                                                                        \
      : "memory")
#define ROMA_RBW_READ_WAIT()                                                                                           \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                  \
               : "+v"(cr[0]), "+v"(cr[1]), "+v"(cr[2]), "+v"(cr[3]), "+v"(cr[4]), "+v"(cr[5]), "+v"(cr[6]), "+v"(cr[7])::"memory")
  ROMA_RBW_WAIT_VM(3 * (NR - 2));  // row 0 has landed: only rows 1 .. NR - 2 may be outstanding
  ROMA_RBW_READ(rd0);

  int slot1 = 1, fill = NR - 1;  // ring slot of input row t + 1; slot the next DMA goes to (= slot of row t - 1)
  // Row t + 1 is read during iteration t, so it must have landed here: at most the DMA issued AFTER it may be outstanding -
  // rows t + 2 .. t + NR - 1, 3 pieces each.  (The younger output stores must not be added to the allowance - a store can
  // retire before an older load: dwconv_ring.hip.)  The output row pointer advances by one image row per finished row.
  char* orow_g = reinterpret_cast<char*>(obase + ((long)ys * W + xw0) * CP);
  const long orow_step = (long)W * CP * 2;
#pragma nounroll
  for (int t = 0; t < T; ++t) {
    ROMA_RBW_ISSUE(t + NR - 1, fill);
    ROMA_RBW_WAIT_VM(3 * (NR - 2));
    const int o = t - 4;  // output row (relative to ys) finished by input row t
    ROMA_RBW_READ_WAIT();
    if (active) {
#define ROMA_RBW_CVT(J)                                                  \
  {                                                                      \
    const uint32_t lo_ = (uint32_t)cr[J], hi_ = (uint32_t)(cr[J] >> 32); \
    v[J][0] = f32x2{h16_lo(lo_), h16_hi(lo_)};                           \
    v[J][1] = f32x2{h16_lo(hi_), h16_hi(hi_)};                           \
  }
      f32x2 v[8][2];
      ROMA_RBW_CVT(0) ROMA_RBW_CVT(1) ROMA_RBW_CVT(2)
      f32x4 wq[2][5];
#pragma unroll
      for (int k = 0; k < 5; ++k) wq[0][k] = *(lds_f32x4*)(wsm + ((4 - k) * 5 + 0) * CP + c);
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        if (kx < 4) {
#pragma unroll
          for (int k = 0; k < 5; ++k) wq[(kx + 1) & 1][k] = *(lds_f32x4*)(wsm + ((4 - k) * 5 + kx + 1) * CP + c);
        }
        ROMA_RBW_CVT(kx + 3)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 5; ++k) {  // acc[k] holds output row t - 4 + k (tap row ky = 4 - k)
          const f32x4 wx = wq[kx & 1][k];
          const f32x2 w0 = f32x2{wx[0], wx[1]}, w1 = f32x2{wx[2], wx[3]};
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            acc[k][px][0] = v[px + kx][0] * w0 + (k == 4 && kx == 0 ? bias0 : acc[k][px][0]);  // (the new row starts from the bias:
            acc[k][px][1] = v[px + kx][1] * w1 + (k == 4 && kx == 0 ? bias1 : acc[k][px][1]);  //  no 16 v_mov per row to re-seed acc[4])
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef ROMA_RBW_CVT
      if (o >= 0) {
        lds_u8* xrow = Xt + xtw0;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          u32x2_t u;
          u.x = pack_relu_h16x2(acc[0][px][0][0], acc[0][px][0][1]);
          u.y = pack_relu_h16x2(acc[0][px][1][0], acc[0][px][1][1]);
          *(lds_u32x2*)(xrow + px * RBW_XROW) = u;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          acc[k][px][0] = acc[k + 1][px][0];
          acc[k][px][1] = acc[k + 1][px][1];
        }
    }
    if (o >= 0) {
      // ---------------- 1x1 convolution of output row o on MFMA, out of the wave's own Xt
      f32x16 oa[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) oa[u][4 * g + j] = g < 3 ? pbias[g][j] : 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const u32x4_t xf = *(lds_u32x4*)(Xt + (u ? xtr1 : xtr0) + ks * 32);
          oa[u] = mfma_h16_32x32x16(wA[ks], xf, oa[u]);
        }
      }
      ROMA_RBW_READ(rd0 + (unsigned)slot1 * RBW_ROWB);  // row t + 1 (past the strip's last row: a stale slot nobody uses)
      if constexpr (FINAL) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pxl = u * 32 + l31;
          // rows 0-2 (lanes 0-31, registers 0-2) + rows 4-6 (lanes 32-63, registers 0-2): head + remainder of the composed map
          const float d0 = oa[u][0] + __shfl_xor(oa[u][0], 32), d1 = oa[u][1] + __shfl_xor(oa[u][1], 32),
                      d2 = oa[u][2] + __shfl_xor(oa[u][2], 32);
          if (hh == 0 && pxl < npw) delta[((long)b * H + ys + o) * W + xw0 + pxl] = f32x4{d0, d1, d2, 0.f};
        }
      } else {
        // A lane holds channels 8 g + 4 hh + [0, 4) of pixel 32 u + l31 (g = 0 .. 2).  v_permlane32_swap(a, b) exchanges lanes
        // 32-63 of a with lanes 0-31 of b: for (g = 0, g = 1) of one block lane (l31, hh) ends up with the 8 consecutive
        // channels 8 hh + [0, 8) of its pixel; for g = 2 of BOTH blocks the lower half-wave gets channels 16 .. 23 of pixel l31,
        // the upper one those of pixel 32 + l31 - i.e. of pixel `lane`.  Three 16-byte stores per lane and row.
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned a0 = pack_bf16x2(oa[u][0], oa[u][1]), a1 = pack_bf16x2(oa[u][2], oa[u][3]);
          const unsigned b0 = pack_bf16x2(oa[u][4], oa[u][5]), b1 = pack_bf16x2(oa[u][6], oa[u][7]);
          const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          const int pxl = u * 32 + l31;
          if (pxl < npw) *reinterpret_cast<u32x4_t*>(orow_g + pxl * (CP * 2) + 16 * hh) = u32x4_t{s0[0], s1[0], s0[1], s1[1]};
        }
        {
          const unsigned a0 = pack_bf16x2(oa[0][8], oa[0][9]), a1 = pack_bf16x2(oa[0][10], oa[0][11]);
          const unsigned b0 = pack_bf16x2(oa[1][8], oa[1][9]), b1 = pack_bf16x2(oa[1][10], oa[1][11]);
          const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          if (lane < npw) *reinterpret_cast<u32x4_t*>(orow_g + lane * (CP * 2) + 32) = u32x4_t{s0[0], s1[0], s0[1], s1[1]};
        }
      }
    } else {
      ROMA_RBW_READ(rd0 + (unsigned)slot1 * RBW_ROWB);
    }
    if (!FINAL && o >= 0) orow_g += orow_step;
    fill = fill + 1 == NR ? 0 : fill + 1;
    slot1 = slot1 + 1 == NR ? 0 : slot1 + 1;
  }
#undef ROMA_RBW_READ_WAIT
#undef ROMA_RBW_READ
#undef ROMA_RBW_ISSUE
  ROMA_RBW_WAIT_VM(0);  // trailing zero-page DMAs must not outlive the workgroup's LDS allocation
}

int g_rb24_wave = -1;  // roma_tuning("rb24w", v): 1 = this kernel for C = 24 (default), 0 = refiner_block_kernel<24>, -1 = env ROMA_RB24W

// 0 = launched, 1 = not this kernel's problem, < 0 = error
int refiner_block24_wave_try_launch(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw, long ldpw,
                                    const float* pw_b, int B, int H, int W, int dt, hipStream_t s, float* delta) {
  static const int env = getenv("ROMA_RB24W") ? atoi(getenv("ROMA_RB24W")) : 1;
#ifdef ROMA_TOOLS_BUILD
  if (!(g_rb24_wave >= 0 ? g_rb24_wave : env)) return 1;  // A/B: the two-barrier workgroup kernel (refiner_block_2b.inc)
#endif
  if (dt != DT_BF16 || H < 1 || W < 1 || (long)H * W * RBW_C * 2 >= (1l << 31)) return 1;  // (32-bit offsets inside an image)
  if ((reinterpret_cast<uintptr_t>(in) & 15) != 0 || (reinterpret_cast<uintptr_t>(delta ? (void*)delta : out) & 15) != 0) return 1;
  if ((reinterpret_cast<uintptr_t>(pw) & 15) != 0 || ldpw % 8 != 0) return 1;
  const int nxg = (W + RBW_PXW - 1) / RBW_PXW;
  // strip height: a strip of SY rows reads SY + 4 input rows and pays ~3 rows of pipeline fill; 2048 waves are resident
  // (8 per CU) and take the tasks in rounds - pick the split of H that minimises rounds x (SY + 7)
  int SY = H;
  {
    long best = -1;
    for (int ns = (H + 95) / 96; ns <= std::max(1, H / 6); ++ns) {
      const int sy = (H + ns - 1) / ns;
      const long nt = (long)B * nxg * ((H + sy - 1) / sy);
      const long cost = ((nt + 2047) / 2048) * (sy + 7);
      if (best < 0 || cost < best) {
        best = cost;
        SY = sy;
      }
    }
  }
  const long ntasks = (long)B * nxg * ((H + SY - 1) / SY);
  ROMA_REQUIRE(ntasks < (1l << 31), "refiner_block: grid too large");
  const long nwg = (ntasks + 3) / 4, wg_per_xcd = (nwg + 7) / 8;  // (the kernel decodes the same arithmetic)
  if (delta)
    hipLaunchKernelGGL(refiner_block24_wave_kernel<true>, dim3((unsigned)(wg_per_xcd * 8)), dim3(256), 0, s, (const bf16_t*)in,
                       (bf16_t*)nullptr, dw_w, dw_b, (const bf16_t*)pw, ldpw, pw_b, B, H, W, SY, nxg, ntasks, (f32x4*)delta);
  else
    hipLaunchKernelGGL(refiner_block24_wave_kernel<false>, dim3((unsigned)(wg_per_xcd * 8)), dim3(256), 0, s, (const bf16_t*)in,
                       (bf16_t*)out, dw_w, dw_b, (const bf16_t*)pw, ldpw, pw_b, B, H, W, SY, nxg, ntasks, (f32x4*)nullptr);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
