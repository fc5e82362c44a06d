// Fused depthwise-5x5 (+folded BN, ReLU) -> 1x1 convolution for the narrow ConvRefiner scales.  See refiner_block.h.
//
// One 256-thread workgroup owns a strip of SY image rows x PX pixels and all CP channels:
//   * depthwise phase: the rolling-window stencil of dwconv5x5_kernel (elementwise.hip) - one thread = 4 channels x
//     4 x-positions, each input row loaded once and fed to the 5 output rows it touches (v_pk_fma_f32), next input
//     row prefetched.  The finished output row (after ReLU) is packed to bf16 into an LDS tile Xt[row][px][k]
//     (80 / 304 byte pixel rows: conflict-free for the 16-byte MFMA fragment reads) instead of HBM;
//   * every R = 2 output rows: barrier, the 4 waves run out[ch][px] = Wpw[ch][:] . Xt[px][:] on
//     v_mfma_f32_32x32x16_bf16 (channels on the MFMA "i" side, so every lane ends up with 4 consecutive channels of
//     one pixel), add the bias, pack to bf16 into a contiguous LDS image of the output rows, barrier, and the whole
//     workgroup streams those rows to HBM with 16-byte lane-contiguous stores (a row segment PX*CP*2 B is contiguous).
//   * C = 144: wave w keeps the weights of output channels [32w, 32w+32) in registers for the whole strip; the
//     16-channel remainder block comes from LDS.  C = 24: the single (padded) 32-channel block comes from LDS.
// The in-flight prefetch of the next input row spans the MFMA phase, which is what hides the HBM latency.
#include "refiner_block.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gemm.h"  // DT_*

namespace roma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// explicit LDS address-space types: 32-bit address arithmetic (generic pointers cost 64-bit VGPR pairs here)
#define ROMA_LDS __attribute__((address_space(3)))
typedef ROMA_LDS unsigned char lds_u8;
typedef ROMA_LDS float lds_f32;
typedef ROMA_LDS f32x4 lds_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;  // native vectors (HIP's u32x4_t is a struct:
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;  //  no address-space-qualified copies)
typedef ROMA_LDS u32x4_t lds_u32x4;
typedef ROMA_LDS u32x2_t lds_u32x2;

__device__ __attribute__((aligned(256))) unsigned int g_rb_zero_page[64];  // source of every out-of-image DMA chunk
__device__ __attribute__((aligned(256))) unsigned int g_rb_dump[512];       // where lanes right of the tile store (C = 144): 64 x 16 B + the largest piece offset

template <int CP> struct RBCfg {
  static constexpr int GC = CP / 4;                        // channel groups of 4
  static constexpr int XQ = CP == 24 ? 36 : 256 / GC;      // x quads per workgroup row   (36 ; 7)
  static constexpr int PX = 4 * XQ;                        // pixels per workgroup row    (144 ; 28)
  static constexpr int PXB = (PX + 31) / 32;               // 32-pixel MFMA blocks / row  (5 ; 1)
  static constexpr int KS = (CP + 15) / 16;                // MFMA k-steps                (2 ; 9)
  static constexpr int KP = KS * 16;                       // padded K                    (32 ; 144)
  static constexpr int NBF = CP / 32;                      // full 32-channel blocks      (0 ; 4)
  static constexpr int TAIL = CP % 32;                     // channels of the last block  (24 ; 16)
  static constexpr int XROW = KP * 2 + 16;                 // bytes per pixel row of Xt   (80 ; 304)
  static constexpr int NR = CP == 24 ? 6 : 4;              // input rows in the LDS ring (DMA runs NR-1 rows ahead)
  static constexpr int IN_ROWB = (PX + 4) * CP * 2;        // bytes of one input row segment incl. halo (7104 ; 9216)
  static constexpr int NDMA = (IN_ROWB + 1023) / 1024;     // 1 KiB DMA instructions per row (7 ; 9)
  static constexpr int RSTRIDE = NDMA * 1024;
  static constexpr int KW = (NDMA + 3) / 4;                // max DMA instructions per wave per row (2 ; 3)
  static constexpr int OROW = PX * CP * 2;                 // bytes of one output row segment (6912 ; 8064)
  static constexpr int ROW16 = OROW / 16;
  static constexpr int OPIX = CP * 2 + (CP >= 128 ? 16 : 0);  // LDS bytes per staged output pixel: 288 -> 304 keeps the 8-byte
                                                                // MFMA-layout writes of 16 pixels on distinct banks (48 is already fine)
  static constexpr int OFF_PWB = 26 * CP * 4;
  static constexpr int OFF_WT = OFF_PWB + CP * 4;
  static constexpr int OFF_XT = OFF_WT + TAIL * XROW;
  static constexpr int OFF_OT = OFF_XT + PXB * 32 * XROW;
  static constexpr int WORK_BYTES = OFF_OT + PX * OPIX;
  static constexpr int RING_BYTES = NR * RSTRIDE;
  static_assert(NBF == 0 || NBF == 4, "one wave per full channel block");
  static_assert(GC * XQ <= 256 && OROW % 16 == 0 && TAIL % 8 == 0 && ROW16 <= 512, "layout");
  static_assert(OFF_PWB % 16 == 0 && OFF_WT % 16 == 0 && OFF_XT % 16 == 0 && OFF_OT % 16 == 0, "alignment");
  static_assert(WORK_BYTES + RING_BYTES <= 80 * 1024, "two workgroups per CU");
  static_assert((NR - 1) * (KW + 2) < 64, "vmcnt range");
};

__device__ __forceinline__ void rb_glds16(const char* src, lds_u8* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (ROMA_LDS void*)lds_wave_base, 16, 0, 0);
}

#define ROMA_RB_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// LDS writes of this wave retired, then the raw barrier (no __syncthreads: its release fence would drain vmcnt and
// with it the whole DMA prefetch queue)
#define ROMA_RB_BARRIER()                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
  __builtin_amdgcn_s_barrier();                             \
  asm volatile("" ::: "memory")

#ifdef ROMA_TOOLS_BUILD
#include "refiner_block_2b.inc"  // refiner_block_kernel<CP>: the two-barrier A/B reference (tools builds only)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// C = 144 with ONE barrier per image row (roma_tuning("rb144_1b", 0) selects the two-barrier kernel of refiner_block_2b.inc in a
// tools build, for A/B).
//
// The two-barrier kernel needs two barriers per row: B2 (the depthwise tile Xt is complete / the ring slot is free) and B3 (the
// output tile Ot is complete before the whole workgroup streams it out / the next input row has landed).  Here
//   * Xt is double buffered: row t's 1x1 reads Xt[t & 1] while the fastest wave may already write Xt[(t + 1) & 1];
//   * every wave stores exactly what it computed - its own 32-channel slice of the row (64 B per pixel) and, when it is its
//     turn, the 16-channel remainder block - so nobody waits for anybody else's part of Ot;
//   * a wave waits for its own DMA pieces of row t + 1 BEFORE the barrier of row t, which makes that barrier the "row t + 1
//     is visible" point as well.
// The second Xt buffer is paid for with the ring: NR = 3 rows (the DMA still runs a full row ahead of the one being
// waited for).  Arithmetic and its order are unchanged: bit-identical results (tests).
constexpr int RB1_XT = RBCfg<144>::PXB * 32 * RBCfg<144>::XROW;
// LDS of this kernel: ring + taps / bias / remainder weights + two Xt buffers (no output tile since round 5)
// (A ring of 4 rows - the LDS the output tile freed - was measured in round 5: 6.68 / 6.73 against 6.71 / 6.78 ms per step, noise;
// the kernel waits for its VALU, not for the DMA.  profiles/r05_v11_block144_ring_depth.log)
constexpr int RB1_NR = 3;
static_assert(RB1_NR * RBCfg<144>::RSTRIDE + RBCfg<144>::OFF_XT + 2 * RB1_XT <= 80 * 1024, "two workgroups per CU");

// FINAL (round 5): the last block of a ConvRefiner, its 1x1 composed with out_conv (see refiner_block24w.hip): `pw` holds 8 rows
// (rows 0-2 head, rows 4-6 remainder of the composed C -> 3 weights), `pwb` the composed bias in [0, 3); the wave whose turn it
// is (o & 3) runs ONE chain of 9 MFMAs per row and writes delta[pixel] = {d flow x, d flow y, d certainty, 0} - no Ot, no
// 288-byte row stores, no remainder block.
template <bool FINAL>
__global__ __launch_bounds__(256, 2) void refiner_block144_1b_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                                     const float* __restrict__ dww, const float* __restrict__ dwb,
                                                                     const bf16_t* __restrict__ pw, long ldpw,
                                                                     const float* __restrict__ pwb, int B, int H, int W, int SY,
                                                                     int nxg, int nblocks, f32x4* __restrict__ delta) {
  constexpr int CP = 144;
  typedef RBCfg<CP> Cf;
  constexpr int GC = Cf::GC, XQ = Cf::XQ, PX = Cf::PX, KS = Cf::KS, NR = RB1_NR, KW = Cf::KW;
  constexpr int XROW = Cf::XROW, NDMA = Cf::NDMA, RSTRIDE = Cf::RSTRIDE;
  static_assert(Cf::PXB == 1 && Cf::NBF == 4 && Cf::TAIL == 16, "one 32-pixel block, four full channel blocks + 16");
  __shared__ __attribute__((aligned(1024))) unsigned char ring[NR * RSTRIDE];
  __shared__ __attribute__((aligned(16))) unsigned char work[Cf::OFF_XT];          // taps, 1x1 bias, remainder weights
  __shared__ __attribute__((aligned(16))) unsigned char xtb[2 * RB1_XT];           // Xt, double buffered
  lds_u8* const wk = (lds_u8*)work;
  lds_f32* const wsm = (lds_f32*)wk;
  lds_f32* const pbs = (lds_f32*)(wk + Cf::OFF_PWB);
  lds_u8* const Wt = wk + Cf::OFF_WT;

  const int per_xcd = (nblocks + 7) / 8;
  const long lb = (long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lb >= nblocks) return;
  const int xg = (int)(lb % nxg);
  long rr = lb / nxg;
  const int yt = (H + SY - 1) / SY;
  const int ys = (int)(rr % yt) * SY;
  const int b = (int)(rr / yt);
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform in a form the compiler sees (round 6: as a plain shift of
                                                              // tid it put the DMA issue - the LDS target depends on wv - behind exec masks)
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int x0 = xg * PX;
  const int sy = min(SY, H - ys);
  const int T = sy + 4;
  const int npx = min(PX, W - x0);

  {
    // taps + bias, 25 + 1 rows of 36 channel groups (16 B each), in channel order
    constexpr int nvec = 26 * GC;
#pragma unroll
    for (int it = 0; it < (nvec + 255) / 256; ++it) {
      const int i = tid + 256 * it;
      if (i < nvec) *(lds_f32x4*)(wsm + i * 4) = *reinterpret_cast<const f32x4*>(i < 25 * GC ? dww + (long)i * 4 : dwb + (long)(i - 25 * GC) * 4);
    }
    if (tid < GC) *(lds_f32x4*)(pbs + tid * 4) = *reinterpret_cast<const f32x4*>(pwb + tid * 4);
    constexpr int wslots = FINAL ? 0 : Cf::TAIL * (XROW / 16);  // (FINAL: `pw` has 8 rows only, and no remainder block)
    for (int i = tid; i < wslots; i += 256) {
      const int n = i / (XROW / 16), sl = i - n * (XROW / 16);
      u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
      if (sl < CP * 2 / 16) v = *reinterpret_cast<const u32x4_t*>(pw + (long)(32 * Cf::NBF + n) * ldpw + sl * 8);
      *(lds_u32x4*)(Wt + n * XROW + sl * 16) = v;
    }
    for (int i = tid; i < 2 * RB1_XT / 16; i += 256) *(lds_u32x4*)((lds_u8*)xtb + i * 16) = u32x4_t{0u, 0u, 0u, 0u};
  }
  u32x4_t wown[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if constexpr (FINAL) {
      wown[ks] = u32x4_t{0u, 0u, 0u, 0u};
      if (l31 < 8) wown[ks] = *reinterpret_cast<const u32x4_t*>(pw + (long)l31 * ldpw + ks * 16 + hh * 8);
    } else {
      wown[ks] = *reinterpret_cast<const u32x4_t*>(pw + (long)(32 * wv + l31) * ldpw + ks * 16 + hh * 8);
    }
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wown[ks]));

  const char* zsrc = reinterpret_cast<const char*>(g_rb_zero_page);
  const char* inb = reinterpret_cast<const char*>(in + ((long)b * H * W) * CP);
  // Round 6 (dwconv_ring.hip): the DMA goes through a buffer descriptor over image b - a 32-bit lane offset fixed for the strip
  // (0x80000000 for chunks past the row segment and columns outside the image: beyond num_records, the hardware returns zeros)
  // plus the row offset in an SGPR; rows above / below the image take the marker behind a wave-uniform branch.  No per-row
  // selects and no 64-bit lane addresses.
  const long pitch = (long)W * CP * 2;  // (H * pitch < 2^31: checked by the launcher)
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(inb), 0, (int)((long)H * pitch), 0x00020000);
  unsigned voff[KW];
#pragma unroll
  for (int q = 0; q < KW; ++q) {
    const int chunk = (wv + 4 * q) * 64 + lane;
    const int x = x0 - 2 + chunk / (CP / 8);
    const bool ok = chunk * 16 < Cf::IN_ROWB && x >= 0 && x < W;
    voff[q] = ok ? (unsigned)((x0 - 2) * CP * 2 + chunk * 16) : 0x80000000u;
  }
  unsigned vout = 0x80000000u;
  asm volatile("" : "+v"(vout));
  const int kw = (NDMA - wv + 3) / 4;  // pieces this wave really issues per row (3 ; 2 ; 2 ; 2)
#define ROMA_RB1_BL16(VOFF, SOFF, DST) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (ROMA_LDS void*)(DST), 16, (int)(VOFF), (int)(SOFF), 0, 0)
#define ROMA_RB1_ISSUE_ROW(RROW, SLOT)                                                                 \
  {                                                                                                    \
    const int yy_ = ys - 2 + (RROW);                                                                   \
    if ((RROW) < T && yy_ >= 0 && yy_ < H) { /* wave-uniform */                                        \
      const int so_ = yy_ * (int)pitch;                                                                \
      _Pragma("unroll") for (int q = 0; q < KW; ++q)                                                   \
        if (wv + 4 * q < NDMA) ROMA_RB1_BL16(voff[q], so_, (lds_u8*)ring + (SLOT) * RSTRIDE + (wv + 4 * q) * 1024); \
    } else {                                                                                           \
      _Pragma("unroll") for (int q = 0; q < KW; ++q)                                                   \
        if (wv + 4 * q < NDMA) ROMA_RB1_BL16(vout, 0, (lds_u8*)ring + (SLOT) * RSTRIDE + (wv + 4 * q) * 1024); \
    }                                                                                                  \
  }

  ROMA_RB_BARRIER();  // staged tiles visible; nothing of ours in flight yet
#pragma unroll
  for (int r = 0; r < NR; ++r) ROMA_RB1_ISSUE_ROW(r, r);
  if (kw == KW) ROMA_RB_WAIT_VM((NR - 1) * KW); else ROMA_RB_WAIT_VM((NR - 1) * (KW - 1));
  ROMA_RB_BARRIER();  // input row 0 landed

  // Lane -> (column quad xq, channel group cg).  Three LDS access streams depend on it, each served in its own lane groups
  // (MI355X_MICROARCH.md, LDS table): the 8 ring reads of a row (ds_read_b64: lanes 0-31 | 32-63, 64 banks), the 25 tap reads
  // (ds_read_b128: {0-3, 12-15, 20-27} {4-11, 16-19, 28-31} and the same + 32, 64 banks) and the 4 Xt writes (ds_write_b64:
  // four groups of 16 consecutive lanes, 32 banks).  A group is free of conflicts when its addresses are distinct modulo
  // the bank row, or equal.  With cg = 16 a + b (a = 0, 1; a = 2 for the four groups 32 .. 35):
  //   ring  byte (4 xq + j) 288 + 8 cg  -> 8-byte unit (16 [xq odd] + cg) mod 32 = 16 (a + [xq odd] mod 2) + b
  //   taps  byte 16 cg                  -> 16-byte unit cg mod 16 = b
  //   Xt    byte (4 xq + px) 304 + 8 cg -> 8-byte unit (8 [xq odd] + cg) mod 16 = b (+ 8)
  // so a BLOCK = the 16 groups of one quad with one a, laid on 16 consecutive lanes in the order of b, is conflict free in
  // all three, and two blocks form a conflict-free half-wave when they are (even quad, a) + (odd quad, a) - complementary
  // ring halves, the same tap addresses - or (quad, a = 0) + (same quad, a = 1):
  //   waves 0-2: lanes 0-15 (2w, a=0) | 16-31 (2w+1, a=0) | 32-47 (2w, a=1) | 48-63 (2w+1, a=1)
  //   wave 3:    lanes 0-31 quad 6, cg = lane; lanes 32-59 the groups 32 .. 35 of the quads 0 2 1 3 | 4 6 5 (4 lanes each)
  // The 28 left-over items (7 quads x groups 32 .. 35) cannot be conflict free: over the workgroup 11 items share each ring
  // unit 0 .. 3 against 8 half-waves, 18 each Xt unit 0 .. 3 against 16 groups.  Gathered in wave 3's upper half they cost 3
  // extra cycles on a ring read and 1 + 1 on an Xt write (the minimum), nothing on a tap read (four addresses, broadcast).
  // History: tid = 36 xq + cg had 2-way conflicts on every tap read (share 0.32, r03_pmc_sq_summary.json); round 3's slot
  // rotation removed them from two of the four b128 groups only, because it assumed 16 CONSECUTIVE lanes per pass (0.229).
  // Pure relabelling of which lane computes which (quad, group): same values, bit for bit.
  int xq, cg;
  if (wv < 3) {
    xq = 2 * wv + ((lane >> 4) & 1);
    cg = (lane & 15) + 16 * (lane >> 5);
  } else if (lane < 32) {
    xq = 6;
    cg = lane;
  } else {
    const int k = lane - 32;
    xq = (0x55643120u >> (4 * (k >> 2))) & 7;  // 0 2 1 3 | 4 6 5 (5: idle lanes 60-63, the addresses of their group)
    cg = 32 + (k & 3);
  }
  const int xb = x0 + xq * 4;
  const bool active = (wv < 3 || lane < 60) && xb < W;
  const int c = cg * 4;        // channel offset: ring reads, Xt writes
  const int cw = cg * 4;       // position inside a tap row of wsm (channel order)
  const f32x4 bx = *(lds_f32x4*)(wsm + 25 * CP + cw);
  const f32x2 bias0 = f32x2{bx[0], bx[1]}, bias1 = f32x2{bx[2], bx[3]};
  f32x2 acc[5][4][2];
#pragma unroll
  for (int s5 = 0; s5 < 5; ++s5)
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      acc[s5][px][0] = bias0;
      acc[s5][px][1] = bias1;
    }
  const unsigned ring_lds = (unsigned)(size_t)((lds_u8*)ring);
  const unsigned rd0 = ring_lds + (unsigned)((xq * 4 * CP + c) * 2);
  bf16_t* obase = out + ((long)b * H * W) * CP;

  int slot = 0;
#pragma nounroll
  for (int t = 0; t < T; ++t) {
    const int o = t - 4;
    lds_u8* const Xt = (lds_u8*)xtb + (t & 1) * RB1_XT;
    if (active) {
      unsigned long long cr[8];
      const unsigned ra = rd0 + (unsigned)slot * RSTRIDE;
      asm volatile(
          "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:%9\n\tds_read_b64 %2, %8 offset:%10\n\t"
          "ds_read_b64 %3, %8 offset:%11\n\tds_read_b64 %4, %8 offset:%12\n\tds_read_b64 %5, %8 offset:%13\n\t"
          "ds_read_b64 %6, %8 offset:%14\n\tds_read_b64 %7, %8 offset:%15\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(cr[0]), "=&v"(cr[1]), "=&v"(cr[2]), "=&v"(cr[3]), "=&v"(cr[4]), "=&v"(cr[5]), "=&v"(cr[6]), "=&v"(cr[7])
          : "v"(ra), "n"(CP * 2), "n"(CP * 4), "n"(CP * 6), "n"(CP * 8), "n"(CP * 10), "n"(CP * 12), "n"(CP * 14)
          : "memory");
#define ROMA_RB_CVT(J)                                                   \
  {                                                                      \
    const uint32_t lo_ = (uint32_t)cr[J], hi_ = (uint32_t)(cr[J] >> 32); \
    v[J][0] = f32x2{h16_lo(lo_), h16_hi(lo_)};                           \
    v[J][1] = f32x2{h16_lo(hi_), h16_hi(hi_)};                           \
  }
      f32x2 v[8][2];
      ROMA_RB_CVT(0) ROMA_RB_CVT(1) ROMA_RB_CVT(2)
      f32x4 wq[2][5];
#pragma unroll
      for (int k = 0; k < 5; ++k) wq[0][k] = *(lds_f32x4*)(wsm + ((4 - k) * 5 + 0) * CP + cw);
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        if (kx < 4) {
#pragma unroll
          for (int k = 0; k < 5; ++k) wq[(kx + 1) & 1][k] = *(lds_f32x4*)(wsm + ((4 - k) * 5 + kx + 1) * CP + cw);
        }
        ROMA_RB_CVT(kx + 3)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
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
#undef ROMA_RB_CVT
      if (o >= 0) {
        lds_u8* xrow = Xt + (xq * 4) * XROW + cg * 8;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          u32x2_t u;
          u.x = pack_relu_h16x2(acc[0][px][0][0], acc[0][px][0][1]);
          u.y = pack_relu_h16x2(acc[0][px][1][0], acc[0][px][1][1]);
          *(lds_u32x2*)(xrow + px * XROW) = u;
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
    // this wave's pieces of input row t + 1 must have landed: the DMA issued after them may stay in flight - row t + 2 only
    // (row t + NR is issued below).  The stores of the last iterations are NOT part of the allowance (a store can retire
    // before an older load: dwconv_ring.hip); they were issued a whole stencil ago.
    if (kw == KW) ROMA_RB_WAIT_VM((NR - 2) * KW); else ROMA_RB_WAIT_VM((NR - 2) * (KW - 1));
    ROMA_RB_BARRIER();  // Xt[t & 1] complete; every wave is done with ring slot `slot`; input row t + 1 is visible
    // (row t + NR is requested at the END of the iteration, behind this row's stores: the counted wait above then leaves exactly
    //  the youngest vector-memory instructions - that row's pieces, loads - in flight, the form tools/audit_vmcnt.py can prove)
    if (o >= 0) {
      // ---------------- 1x1 convolution of output row o: own 32-channel block (weights in registers) ...
      int lanev = lane;  // opaque copy: keeps loop-invariant addresses from being hoisted into long-lived registers
      asm volatile("" : "+v"(lanev));
      const int l31v = lanev & 31, hhv = lanev >> 5;
      if constexpr (FINAL) {
        if ((o & 3) == wv) {  // wave-uniform: the rows of a strip take turns
          f32x16 oa;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 bq = {0.f, 0.f, 0.f, 0.f};
            if (g == 0 && hhv == 0) bq = *(lds_f32x4*)(pbs);  // composed bias in rows 0-2 (row 3 is zero)
#pragma unroll
            for (int j = 0; j < 4; ++j) oa[4 * g + j] = bq[j];
          }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const u32x4_t xf = *(lds_u32x4*)(Xt + l31v * XROW + ks * 32 + hhv * 16);
            oa = mfma_h16_32x32x16(wown[ks], xf, oa);
          }
          const float d0 = oa[0] + __shfl_xor(oa[0], 32), d1 = oa[1] + __shfl_xor(oa[1], 32), d2 = oa[2] + __shfl_xor(oa[2], 32);
          if (hhv == 0 && l31v < npx) delta[((long)b * H + ys + o) * W + x0 + l31v] = f32x4{d0, d1, d2, 0.f};
        }
      } else {
      // this lane's store position: pixel l31, byte 64 wv + 16 hh of it (lanes right of the tile: their slot of the dump page)
      char* const orow_w = reinterpret_cast<char*>(obase + ((long)(ys + o) * W + x0) * CP) + 64 * wv;
      char* const pl = l31v < npx ? orow_w + l31v * (CP * 2) + 16 * hhv : reinterpret_cast<char*>(g_rb_dump) + lanev * 16;
      u32x4_t q_first;
      {
        f32x16 oa;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bq = *(lds_f32x4*)(pbs + 32 * wv + 8 * g + 4 * hhv);
#pragma unroll
          for (int j = 0; j < 4; ++j) oa[4 * g + j] = bq[j];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const u32x4_t xf = *(lds_u32x4*)(Xt + l31v * XROW + ks * 32 + hhv * 16);
          oa = mfma_h16_32x32x16(wown[ks], xf, oa);
        }
        // A lane holds channels 32 wv + 8 g + 4 hh + [0, 4) of pixel l31; v_permlane32_swap pairs the half-waves so that lane
        // (l31, hh) ends up with the 8 consecutive channels 32 wv + 16 P + 8 hh + [0, 8): two 16-byte stores, 32 contiguous
        // bytes per pixel and instruction, straight from the accumulators (until round 5 the row went through an LDS tile
        // Ot: 4 ds_write_b64 + 3 ds_read_b128 + a wait per row, 2-way bank conflicts on both sides).  Lanes right of the
        // tile store into a dump page: exactly three store instructions per wave and row, whatever the lanes' validity.
#pragma unroll
        for (int P = 0; P < 2; ++P) {
          const unsigned a0 = pack_bf16x2(oa[8 * P + 0], oa[8 * P + 1]), a1 = pack_bf16x2(oa[8 * P + 2], oa[8 * P + 3]);
          const unsigned b0 = pack_bf16x2(oa[8 * P + 4], oa[8 * P + 5]), b1 = pack_bf16x2(oa[8 * P + 6], oa[8 * P + 7]);
          const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          const u32x4_t q = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4_t*>(pl + 32 * P) = q;
          if (P == 0) q_first = q;
        }
      }
      // ... and, when it is this wave's turn, the 16-channel remainder block (weights from LDS)
      const bool mine = ((o & 3) == wv);  // wave-uniform
      if (mine) {
        const int wrow = l31v < Cf::TAIL ? l31v : l31v - Cf::TAIL;
        f32x16 ta;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 bq = {0.f, 0.f, 0.f, 0.f};
          if (8 * g + 4 * hhv < Cf::TAIL) bq = *(lds_f32x4*)(pbs + 32 * Cf::NBF + 8 * g + 4 * hhv);
#pragma unroll
          for (int j = 0; j < 4; ++j) ta[4 * g + j] = bq[j];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const u32x4_t wf = *(lds_u32x4*)(Wt + wrow * XROW + ks * 32 + hhv * 16);
          const u32x4_t xf = *(lds_u32x4*)(Xt + l31v * XROW + ks * 32 + hhv * 16);
          ta = mfma_h16_32x32x16(wf, xf, ta);
        }
        // 16 channels: g = 0, 1 hold them (lane: 8 g + 4 hh + [0, 4)); after the swap lane (l31, hh) has 128 + 8 hh + [0, 8)
        const unsigned a0 = pack_bf16x2(ta[0], ta[1]), a1 = pack_bf16x2(ta[2], ta[3]);
        const unsigned b0 = pack_bf16x2(ta[4], ta[5]), b1 = pack_bf16x2(ta[6], ta[7]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4_t q = {s0[0], s1[0], s0[1], s1[1]};
        *reinterpret_cast<u32x4_t*>(pl + 64 * (Cf::NBF - wv)) = q;
      } else {
        // not this wave's turn: its first piece once more, to keep the store count per row uniform for the counted waits
        *reinterpret_cast<u32x4_t*>(pl) = q_first;
      }
    }
      }
    ROMA_RB1_ISSUE_ROW(t + NR, slot);
    slot = slot + 1 == NR ? 0 : slot + 1;
  }
  ROMA_RB_WAIT_VM(0);  // trailing zero-page DMAs must not outlive the workgroup's LDS allocation
}
#undef ROMA_RB1_ISSUE_ROW

// (A stand-alone depthwise kernel on the same LDS-DMA ring was measured for the wide scales, C = 576 / 1152 / 1408:
//  0.541 vs 0.504 ms at 16x216x216x576, 0.299 vs 0.266 ms at 16x108x108x1152 - slower than the register-prefetch
//  kernel in elementwise.hip.  The depthwise phase is bound by its 200 v_pk_fma_f32 + 25 LDS weight reads per row,
//  not by load latency; the ring only pays off here, where it also frees the registers the 1x1 weights need.)
int g_rb144_1b = -1;  // roma_tuning("rb144_1b", v): 1 = refiner_block144_1b_kernel (default), 0 = refiner_block_kernel<144>, -1 = env ROMA_RB144_1B
bool refiner_block_supported(int Cp, int dt) { return dt == DT_BF16 && (Cp == 24 || Cp == 144); }

template <int CP>
static int launch_cp(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw, long ldpw,
                     const float* pw_b, int B, int H, int W, hipStream_t s, float* delta = nullptr) {
  typedef RBCfg<CP> Cf;
  ROMA_REQUIRE((long)H * W * CP * 2 < (1l << 31), "refiner_block: an image must stay below 2 GiB (32-bit offsets inside its buffer descriptor)");
  const int nxg = (W + Cf::PX - 1) / Cf::PX;
  // strip height: SY + 4 input rows are read per strip (and ~2 more rows' worth of pipeline fill), and the 512 resident
  // workgroups take the strips in rounds - pick the split of H that minimises rounds x (SY + 6).  ROMA_RB_SY overrides.
  static const int sy_env = getenv("ROMA_RB_SY") ? atoi(getenv("ROMA_RB_SY")) : 0;
  int SY = sy_env;
  if (SY <= 0) {
    long best = -1;
    for (int yt = 1; yt <= std::max(1, H / 8); ++yt) {
      const int sy = (H + yt - 1) / yt;
      const long n = (long)B * nxg * ((H + sy - 1) / sy);
      const long cost = ((n + 511) / 512) * (sy + 6);
      if (best < 0 || cost < best) {
        best = cost;
        SY = sy;
      }
    }
  }
  const long nb = (long)B * ((H + SY - 1) / SY) * nxg;
  ROMA_REQUIRE(nb < (1l << 30), "refiner_block: grid too large");
  const int nblocks = (int)nb;
  dim3 grid((unsigned)(((nblocks + 7) / 8) * 8));
#ifdef ROMA_TOOLS_BUILD
  static const int dbg = getenv("ROMA_RB_DBG") ? atoi(getenv("ROMA_RB_DBG")) : 0;  // tuning ablations only
  if (dbg & 16) {
    int nb_cu = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_cu, refiner_block_kernel<CP>, 256, 0);
    fprintf(stderr, "refiner_block<%d>: %d workgroups/CU, grid %d\n", CP, nb_cu, nblocks);
  }
  static const int env1b = getenv("ROMA_RB144_1B") ? atoi(getenv("ROMA_RB144_1B")) : 1;
  if (!delta && (CP == 24 || !(g_rb144_1b >= 0 ? g_rb144_1b : env1b) || dbg)) {
    hipLaunchKernelGGL(refiner_block_kernel<CP>, grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, dw_w, dw_b,
                       (const bf16_t*)pw, ldpw, pw_b, B, H, W, SY, nxg, nblocks, dbg);
    ROMA_LAUNCH_CHECK();
    return 0;
  }
#else
  ROMA_REQUIRE(CP == 144, "refiner_block: C = 24 runs on the wave-private kernel only (it declined these tensors: 16-byte aligned bf16 in / out / weights are required)");
  ROMA_REQUIRE(g_rb144_1b != 0, "refiner_block: the two-barrier A/B kernel is not part of this build (make TOOLS=1)");
#endif
  if (delta)
    hipLaunchKernelGGL(refiner_block144_1b_kernel<true>, grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)nullptr, dw_w, dw_b,
                       (const bf16_t*)pw, ldpw, pw_b, B, H, W, SY, nxg, nblocks, (f32x4*)delta);
  else
    hipLaunchKernelGGL(refiner_block144_1b_kernel<false>, grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, dw_w, dw_b,
                       (const bf16_t*)pw, ldpw, pw_b, B, H, W, SY, nxg, nblocks, (f32x4*)nullptr);
  ROMA_LAUNCH_CHECK();
  return 0;
}

int refiner_block_launch(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw, long ldpw,
                         const float* pw_b, int B, int H, int W, int Cp, int dt, hipStream_t s) {
  ROMA_REQUIRE(refiner_block_supported(Cp, dt), "refiner_block: only bf16 with Cp = 24 or 144 is fused");
  ROMA_REQUIRE(in != out, "refiner_block: in and out must not alias");
  ROMA_REQUIRE(ldpw % 8 == 0, "refiner_block: 1x1 weight rows must be 16-byte aligned");
  ProfScope ps(Cp == 24 ? "refiner_block_kernel<24>" : "refiner_block_kernel<144>", 2.0 * (double)B * H * W * Cp * 2.0,
               "byte", s);
  if (Cp == 24) {
    const int rc = refiner_block24_wave_try_launch(in, out, dw_w, dw_b, pw, ldpw, pw_b, B, H, W, dt, s);
    if (rc <= 0) return rc;
    return launch_cp<24>(in, out, dw_w, dw_b, pw, ldpw, pw_b, B, H, W, s);
  }
  return launch_cp<144>(in, out, dw_w, dw_b, pw, ldpw, pw_b, B, H, W, s);
}

// The last block of a ConvRefiner with its 1x1 composed with out_conv: pw_final [8][ldpw] 16-bit (rows 0-2 head, 4-6 remainder of
// the composed [3][Cp] weights), bias_final f32 [Cp] (composed bias in [0, 3), zeros behind); delta [B*H*W][4] f32 receives
// {d flow x, d flow y, d certainty, 0} per pixel (refiner_apply_delta_launch adds them to the running flow / certainty).
int refiner_block_final_launch(const void* in, float* delta, const float* dw_w, const float* dw_b, const void* pw_final, long ldpw,
                               const float* bias_final, int B, int H, int W, int Cp, int dt, hipStream_t s) {
  ROMA_REQUIRE(refiner_block_supported(Cp, dt), "refiner_block_final: only bf16 with Cp = 24 or 144");
  ROMA_REQUIRE(ldpw % 8 == 0 && (reinterpret_cast<uintptr_t>(delta) & 15) == 0, "refiner_block_final: 16-byte aligned weight rows and delta");
  // algorithmic bytes: the block input once + the 16-byte delta per pixel
  ProfScope ps(Cp == 24 ? "refiner_block_final_kernel<24>" : "refiner_block_final_kernel<144>",
               (double)B * H * W * (Cp * 2.0 + 16.0), "byte", s);
  if (Cp == 24) {
    const int rc = refiner_block24_wave_try_launch(in, nullptr, dw_w, dw_b, pw_final, ldpw, bias_final, B, H, W, dt, s, delta);
    if (rc == 1) {
      set_error("refiner_block_final: the C = 24 kernel declined these tensors (16-byte aligned bf16 input / weights)");
      return -1;
    }
    return rc;
  }
  return launch_cp<144>(in, nullptr, dw_w, dw_b, pw_final, ldpw, bias_final, B, H, W, s, delta);
}

}  // namespace roma
