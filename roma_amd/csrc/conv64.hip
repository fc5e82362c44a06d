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
//   * LDS position (pixel p, chunk c) of a ring row holds source chunk c ^ ((p >> 1) & 7).  Pixels are 128 bytes apart, so
//     two of them share a 256-byte bank row; a ds_read_b128 lane group is 16 lanes whose pixels are distinct mod 16
//     (MI355X_MICROARCH.md, LDS table), and (p & 1, c ^ ((p >> 1) & 7)) gives those 16 pixels 16 different 16-byte slots.
//     (The first version swizzled by p & 7: 8 classes for 16 lanes, every read 2-way conflicted - 720 -> see DESIGN.md.)
//   * output: bias + ReLU + bf16 pack in registers, 8-byte stores (4 channels of one pixel per lane).
#include "conv64.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gemm_device.h"

namespace roma {

static __device__ __attribute__((aligned(256))) unsigned int g_c64_dump[256];  // where lanes right of the image store

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
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];  // [4][RSTRIDE], then the bias

  if ((int)blockIdx.x >= nblocks) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % NCG, pg = wave / NCG;
  const int l31 = lane & 31, h = lane >> 5;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned ring0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)ring);

  // ---- this wave's weights: W[32 cg + l31][k = 16 ks + 8 h .. + 8), ks = tap * 4 + g  -> 36 x 16 bytes per lane
  u32x4 wreg[36];
  {
    const bf16_t* wp = w + (long)(32 * cg + l31) * 576 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 36; ++ks) wreg[ks] = *reinterpret_cast<const u32x4*>(wp + 16 * ks);
  }
  // bias lives in LDS behind the ring (registers are for W): channels 32 cg + 8 rg + 4 h + [0, 4) are read per output row
  float* bias_s = reinterpret_cast<float*>(ring + 4 * RSTRIDE);
  if (tid < COUT) bias_s[tid] = bias[tid];

  // ---- fragment read offsets inside a ring row: pixel p = 64 pg + l31 + dx (+ 32 tm), chunk 2 g + h.
  // offset of chunk (2 g + h) ^ s(p) = (chunk h ^ s(p)) with bits 5-6 flipped by g: one register per dx
  unsigned rdo[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int p = 64 * pg + l31 + dx;
    rdo[dx] = ring0 + (unsigned)(p * 128 + ((h ^ ((p >> 1) & 7)) << 4));
  }

  // ---- persistent over strips (SY rows x TW pixels of one image): the weights are fetched once per workgroup
  for (int lb = blockIdx.x; lb < nblocks; lb += gridDim.x) {
  const int xt = lb % nxt;
  const int r = lb / nxt;
  const int yt = (H + SY - 1) / SY;
  const int ys = (r % yt) * SY;
  const int b = r / yt;
  const int x0 = xt * TW;
  const int sy = min(SY, H - ys);
  if (lb != (int)blockIdx.x) __builtin_amdgcn_s_barrier();  // every wave is past the last ring read of the strip before

  // ---- DMA geometry, fixed for the strip: piece q = wave + 4 j of a ring row, lane -> (pixel q * 8 + lane / 8, 16-byte
  // slot lane % 8).  Lanes whose pixel is outside the image (left / right halo at the image border, padding behind the last
  // pixel) are switched off in every DMA; their ring positions are zeroed once, here, in all four slots.
  const bf16_t* inb = in + (long)b * H * W * 64;
  // (round 6, gemm8p.hip: the row DMA through a buffer descriptor over image b - SGPR base + the lane's 32-bit offset + the row
  //  offset in an SGPR - instead of a 64-bit lane address per piece; H * W * 128 B < 2^31: checked by the launcher)
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(inb), 0, (int)((long)H * W * 128), 0x00020000);
  unsigned voff[KW];  // byte offset of the lane's source chunk inside an image row
  bool okx[KW];
  {
    const int dl_p = lane >> 3, dl_slot = lane & 7;
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      const int q = wave + 4 * j;
      const int p = q * 8 + dl_p;  // ring pixel 0 .. TW + 1 (beyond: padding of the last piece)
      const int x = x0 - 1 + p;
      const int c = dl_slot ^ ((p >> 1) & 7);
      okx[j] = q < NPIECE && p < TW + 2 && x >= 0 && x < W;
      voff[j] = okx[j] ? (unsigned)(x * 128 + c * 16) : 0u;
      if (q < NPIECE && !okx[j]) {
        int ln = lane;  // opaque: the address is computed here, not hoisted to kernel entry and parked in scratch
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
          asm volatile("ds_write_b128 %0, %1" ::"v"(ring0 + sl * RSTRIDE + q * 1024 + ln * 16), "v"(z4) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the ordinary loads are retired before the first DMA is counted

  // one input row (image row YY, may be outside the image: zeros) into ring slot SLOT
#define C64_ISSUE_ROW(YY, SLOT)                                                                                \
  {                                                                                                            \
    const int yy_ = (YY);                                                                                      \
    if (yy_ >= 0 && yy_ < H) {                                                                                 \
      const int so_ = yy_ * W * 128;                                                                           \
      _Pragma("unroll") for (int j = 0; j < KW; ++j) {                                                         \
        const int q_ = wave + 4 * j;                                                                           \
        if (q_ < NPIECE) {                                                                                     \
          if (okx[j])                                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(ring + (SLOT) * RSTRIDE + q_ * 1024), \
                                                     16, (int)voff[j], so_, 0, 0);                             \
        }                                                                                                      \
      }                                                                                                        \
    } else { /* rows above / below the image (first and last strip only) */                                    \
      const u32x4 z4_ = {0u, 0u, 0u, 0u};                                                                      \
      int ln_ = lane;                                                                                          \
      asm volatile("" : "+v"(ln_));                                                                            \
      _Pragma("unroll") for (int j = 0; j < KW; ++j) {                                                         \
        const int q_ = wave + 4 * j;                                                                           \
        if (q_ < NPIECE)                                                                                       \
          asm volatile("ds_write_b128 %0, %1" ::"v"(ring0 + (SLOT) * RSTRIDE + q_ * 1024 + ln_ * 16), "v"(z4_) : "memory"); \
      }                                                                                                        \
    }                                                                                                          \
  }

  u32x4 fa[2][4];  // [tm][g] fragments: group g of the current tap until its MFMAs are issued, then of the next tap
#define C64_READ_G(SLOTOFF, DX, G)                                                                            \
  {                                                                                                           \
    const unsigned ad_ = (rdo[DX] + (SLOTOFF)) ^ (unsigned)((G) << 5);                                         \
    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0][G]) : "v"(ad_));                                           \
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fa[1][G]) : "v"(ad_));                              \
  }
  // the two reads of group G are the oldest of the eight in flight (six younger ones: the rest of this tap, the head of the next)
#define C64_WAIT_G(G, N) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[0][G]), "+v"(fa[1][G]) : "n"(N) : "memory")

  // ---- prologue: input rows ys - 1, ys, ys + 1 (ring slot = (row - ys + 1) & 3)
  C64_ISSUE_ROW(ys - 1, 0)
  C64_ISSUE_ROW(ys, 1)
  C64_ISSUE_ROW(ys + 1, 2)

  bf16_t* outb = out + (long)b * H * W * COUT;
  for (int o = 0; o < sy; ++o) {
    // input row ys + o + 1 (the last of this output row's three) was issued one iteration ago; behind it only the 4 output
    // stores of that iteration, which may NOT stay in flight: a store can retire before an older load, so vmcnt(4) would
    // let a wave through with the row still on its way (dwconv_ring.hip).  First iteration: everything the prologue issued.
    if (o == 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      C64_WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
    C64_ISSUE_ROW(ys + o + 2, (o + 3) & 3)  // into the slot of input row ys + o - 2: every wave is past its last read
    if (o == 0) {  // later rows: the first half of tap 0 was requested behind the last tap of the row before
      C64_READ_G(0, 0, 0)
      C64_READ_G(0, 0, 1)
    }
    C64_READ_G((o & 3) * RSTRIDE, 0, 2)
    C64_READ_G((o & 3) * RSTRIDE, 0, 3)

    f32x16 acc[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][e] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // younger reads in flight behind group g's two: 6 (rest of this tap + head of the next), except for the very last
        // group of the row, behind which only the half tap of the next row was requested (4).  Where fewer are in flight
        // (row start) the wait merely asks for more than it needs.
        if (t == 8 && g == 3) {
          C64_WAIT_G(g, 4);
        } else {
          C64_WAIT_G(g, 6);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          acc[tm] = mfma_h16_32x32x16(wreg[t * 4 + g],
                                                            fa[tm][g], acc[tm]);
        // group g of the next tap overwrites fa[.][g] behind the ISSUED MFMAs (operands are read at issue, LDS data comes
        // back >= 64 cycles later); behind the last tap: tap 0 of the next output row, whose ring row is already resident.
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < 9) {
          C64_READ_G(((o + (t + 1) / 3) & 3) * RSTRIDE, (t + 1) % 3, g)
        } else if (g < 2) {  // (the other half after the epilogue: 16 registers it needs)
          C64_READ_G(((o + 1) & 3) * RSTRIDE, 0, g)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- bias + ReLU + bf16.  A lane holds channels 8 rg + 4 h + [0, 4) of pixel l31; v_permlane32_swap pairs the two
    // halves of a wave so that lane (l31, h) ends up with the 8 consecutive channels 16 P + 8 h + [0, 8): 16-byte stores,
    // 32 contiguous bytes per pixel and instruction.  Exactly 4 store instructions per wave and row, whatever the lanes'
    // validity (the vmcnt above counts on it): lanes right of the image store into a dump buffer.
    const int y = ys + o;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      f32x4 bv0, bv1;
      {
        const unsigned ba = ring0 + 4 * RSTRIDE + (32 * cg + 16 * P + 4 * h) * 4;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(bv0), "=&v"(bv1)
                     : "v"(ba)
                     : "memory");
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int x = x0 + 64 * pg + 32 * tm + l31;
        const unsigned a0 = pack_bf16x2(fmaxf(acc[tm][8 * P + 0] + bv0[0], 0.f), fmaxf(acc[tm][8 * P + 1] + bv0[1], 0.f));
        const unsigned a1 = pack_bf16x2(fmaxf(acc[tm][8 * P + 2] + bv0[2], 0.f), fmaxf(acc[tm][8 * P + 3] + bv0[3], 0.f));
        const unsigned b0 = pack_bf16x2(fmaxf(acc[tm][8 * P + 4] + bv1[0], 0.f), fmaxf(acc[tm][8 * P + 5] + bv1[1], 0.f));
        const unsigned b1 = pack_bf16x2(fmaxf(acc[tm][8 * P + 6] + bv1[2], 0.f), fmaxf(acc[tm][8 * P + 7] + bv1[3], 0.f));
        // swap(a, b): lanes 32-63 of a <-> lanes 0-31 of b.  h = 0 keeps its rg = 2P data and receives the partner's
        // rg = 2P (channels + 4); h = 1 receives the partner's rg = 2P + 1 and keeps its own.
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4 st = {s0[0], s1[0], s0[1], s1[1]};
        bf16_t* op = outb + ((long)y * W + x) * COUT + 32 * cg + 16 * P + 8 * h;
        u32x4* dst = x < W ? reinterpret_cast<u32x4*>(op) : reinterpret_cast<u32x4*>(g_c64_dump) + lane;
        *dst = st;  // (a non-temporal hint here DOUBLES the kernel's time: the 32-byte pieces then reach HBM unmerged)
      }
    }
  }
  C64_WAIT_VM(0);  // trailing DMA (rows below the strip) must neither outlive the workgroup's LDS nor land in the next strip
  }  // strips
#undef C64_WAIT_G
#undef C64_READ_G
#undef C64_ISSUE_ROW
}

// ---------------------------------------------------------------------------------------------------------------------
// Cin = 128, Cout = 128 (VGG conv2_2).  The weights are 288 KiB: exactly the 8 x 144 weight registers of EIGHT waves, so a
// workgroup is 512 threads, wave = (channel quarter cg, Cin half kh), and each wave multiplies its 32 output channels by
// its 64 input channels over all 9 taps (the same 72 MFMAs per output row as above).  The two kh waves of a quarter hold
// partial sums of the same 32 x 64 outputs: each writes the pixel block the OTHER finishes (tm = 1 - kh) into an LDS
// exchange buffer after its last tap; the row's barrier (the one the ring needs anyway) publishes it, and behind the
// barrier each wave adds its partner's block to its own (tm = kh), applies bias + ReLU and stores - i.e. the epilogue of
// row o runs at the top of row o + 1 and needs no barrier of its own.  The exchange buffers alternate with the row parity
// (a wave two taps ahead must not overwrite what its partner has not read).
// Ring: 4 slots x 66 pixels x 256 B; a pixel is one 256-byte bank row, position (p, c) holds source chunk c ^ (p & 15).
template <int DUMMY>
__global__ __launch_bounds__(512, 1) void conv3x3_c128_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ out, int B,
                                                              int H, int W, int SY, int nxt, int nblocks) {
  constexpr int COUT = 128, TW = 64;
  constexpr int NPIECE = ((TW + 2) * 256 + 1023) / 1024;  // 17
  constexpr int RSTRIDE = NPIECE * 1024;
  constexpr int KW = (NPIECE + 7) / 8;  // 3
  constexpr int XCH = 4 * RSTRIDE;      // exchange buffers: [parity][wave][4 KiB]
  constexpr int BIAS = XCH + 2 * 8 * 4096;
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  if ((int)blockIdx.x >= nblocks) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave & 3, kh = wave >> 2;
  const int l31 = lane & 31, h = lane >> 5;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned ring0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)ring);

  // W[32 cg + l31][k = tap * 128 + 64 kh + 16 g + 8 h .. + 8)
  u32x4 wreg[36];
  {
    const bf16_t* wp = w + (long)(32 * cg + l31) * 1152 + 64 * kh + 8 * h;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) wreg[t * 4 + g] = *reinterpret_cast<const u32x4*>(wp + 128 * t + 16 * g);
  }
  float* bias_s = reinterpret_cast<float*>(ring + BIAS);
  if (tid < COUT) bias_s[tid] = bias[tid];

  unsigned rdo[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int p = l31 + dx;
    rdo[dx] = ring0 + (unsigned)(p * 256 + (((8 * kh + h) ^ (p & 15)) << 4));
  }
  // exchange addresses: lane-contiguous 16-byte pieces, 4 per lane (j * 1 KiB apart)
  const unsigned xw = ring0 + XCH + (unsigned)((cg + 4 * (1 - kh)) * 4096 + lane * 16);  // what the partner will finish
  const unsigned xr = ring0 + XCH + (unsigned)((cg + 4 * kh) * 4096 + lane * 16);        // what the partner left for this wave

  for (int lb = blockIdx.x; lb < nblocks; lb += gridDim.x) {
  const int xt = lb % nxt;
  const int r = lb / nxt;
  const int yt = (H + SY - 1) / SY;
  const int ys = (r % yt) * SY;
  const int b = r / yt;
  const int x0 = xt * TW;
  const int sy = min(SY, H - ys);
  if (lb != (int)blockIdx.x) __builtin_amdgcn_s_barrier();

  const bf16_t* inb = in + (long)b * H * W * 128;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(inb), 0, (int)((long)H * W * 256), 0x00020000);  // (conv3x3_c64_kernel)
  unsigned voff[KW];
  bool okx[KW];
  {
    const int dl_p = lane >> 4, dl_slot = lane & 15;
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      const int q = wave + 8 * j;
      const int p = q * 4 + dl_p;
      const int x = x0 - 1 + p;
      const int c = dl_slot ^ (p & 15);
      okx[j] = q < NPIECE && p < TW + 2 && x >= 0 && x < W;
      voff[j] = okx[j] ? (unsigned)(x * 256 + c * 16) : 0u;
      if (q < NPIECE && !okx[j]) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
          asm volatile("ds_write_b128 %0, %1" ::"v"(ring0 + sl * RSTRIDE + q * 1024 + ln * 16), "v"(z4) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

#define C128_ISSUE_ROW(YY, SLOT)                                                                               \
  {                                                                                                            \
    const int yy_ = (YY);                                                                                      \
    if (yy_ >= 0 && yy_ < H) {                                                                                 \
      const int so_ = yy_ * W * 256;                                                                           \
      _Pragma("unroll") for (int j = 0; j < KW; ++j) {                                                         \
        const int q_ = wave + 8 * j;                                                                           \
        if (q_ < NPIECE) {                                                                                     \
          if (okx[j])                                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(ring + (SLOT) * RSTRIDE + q_ * 1024), \
                                                     16, (int)voff[j], so_, 0, 0);                             \
        }                                                                                                      \
      }                                                                                                        \
    } else {                                                                                                   \
      const u32x4 z4_ = {0u, 0u, 0u, 0u};                                                                      \
      int ln_ = lane;                                                                                          \
      asm volatile("" : "+v"(ln_));                                                                            \
      _Pragma("unroll") for (int j = 0; j < KW; ++j) {                                                         \
        const int q_ = wave + 8 * j;                                                                           \
        if (q_ < NPIECE)                                                                                       \
          asm volatile("ds_write_b128 %0, %1" ::"v"(ring0 + (SLOT) * RSTRIDE + q_ * 1024 + ln_ * 16), "v"(z4_) : "memory"); \
      }                                                                                                        \
    }                                                                                                          \
  }

  u32x4 fa[2][4];
#define C128_READ_G(SLOTOFF, DX, G)                                                                           \
  {                                                                                                           \
    const unsigned ad_ = (rdo[DX] + (SLOTOFF)) ^ (unsigned)((G) << 5);                                         \
    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0][G]) : "v"(ad_));                                           \
    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fa[1][G]) : "v"(ad_));                              \
  }
#define C128_WAIT_G(G, N) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[0][G]), "+v"(fa[1][G]) : "n"(N) : "memory")

  // finish output row Y from this wave's block ACC and the partner's partial sums in exchange buffer PAR
  bf16_t* outb = out + (long)b * H * W * COUT;
#define C128_EPILOGUE(ACC, Y, PAR)                                                                            \
  {                                                                                                           \
    f32x4 pp_[4];                                                                                             \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t" \
                 "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"                                    \
                 : "=&v"(pp_[0]), "=&v"(pp_[1]), "=&v"(pp_[2]), "=&v"(pp_[3])                                 \
                 : "v"(xr + (unsigned)(PAR) * 32768u)                                                         \
                 : "memory");                                                                                 \
    const int x_ = x0 + 32 * kh + l31;                                                                        \
    _Pragma("unroll") for (int P = 0; P < 2; ++P) {                                                           \
      f32x4 bv0_, bv1_;                                                                                       \
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)"            \
                   : "=&v"(bv0_), "=&v"(bv1_)                                                                 \
                   : "v"(ring0 + BIAS + (32 * cg + 16 * P + 4 * h) * 4)                                       \
                   : "memory");                                                                               \
      const f32x4 p0_ = pp_[2 * P], p1_ = pp_[2 * P + 1];                                                     \
      const unsigned a0_ = pack_bf16x2(fmaxf(ACC[8 * P + 0] + p0_[0] + bv0_[0], 0.f), fmaxf(ACC[8 * P + 1] + p0_[1] + bv0_[1], 0.f)); \
      const unsigned a1_ = pack_bf16x2(fmaxf(ACC[8 * P + 2] + p0_[2] + bv0_[2], 0.f), fmaxf(ACC[8 * P + 3] + p0_[3] + bv0_[3], 0.f)); \
      const unsigned b0_ = pack_bf16x2(fmaxf(ACC[8 * P + 4] + p1_[0] + bv1_[0], 0.f), fmaxf(ACC[8 * P + 5] + p1_[1] + bv1_[1], 0.f)); \
      const unsigned b1_ = pack_bf16x2(fmaxf(ACC[8 * P + 6] + p1_[2] + bv1_[2], 0.f), fmaxf(ACC[8 * P + 7] + p1_[3] + bv1_[3], 0.f)); \
      const auto s0_ = __builtin_amdgcn_permlane32_swap(a0_, b0_, false, false);                              \
      const auto s1_ = __builtin_amdgcn_permlane32_swap(a1_, b1_, false, false);                              \
      const u32x4 st_ = {s0_[0], s1_[0], s0_[1], s1_[1]};                                                     \
      bf16_t* op_ = outb + ((long)(Y) * W + x_) * COUT + 32 * cg + 16 * P + 8 * h;                            \
      u32x4* dst_ = x_ < W ? reinterpret_cast<u32x4*>(op_) : reinterpret_cast<u32x4*>(g_c64_dump) + lane;     \
      *dst_ = st_;                                                                                            \
    }                                                                                                         \
  }
  // hand the block the partner finishes to the exchange buffer of parity PAR
#define C128_PUBLISH(ACC, PAR)                                                                                \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
    const f32x4 v_ = {ACC[4 * j + 0], ACC[4 * j + 1], ACC[4 * j + 2], ACC[4 * j + 3]};                        \
    asm volatile("ds_write_b128 %0, %1" ::"v"(xw + (unsigned)(PAR) * 32768u + j * 1024), "v"(v_) : "memory"); \
  }

  C128_ISSUE_ROW(ys - 1, 0)
  C128_ISSUE_ROW(ys, 1)
  C128_ISSUE_ROW(ys + 1, 2)

  f32x16 acc[2];
  for (int o = 0; o < sy; ++o) {
    // the row DMA issued one iteration ago is the youngest memory operation of this wave (the stores of the epilogue
    // went out before it); the LDS writes of the exchange are retired before the barrier publishes them
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (o > 0) {
      if (kh == 0) C128_EPILOGUE(acc[0], ys + o - 1, (o - 1) & 1)
      else C128_EPILOGUE(acc[1], ys + o - 1, (o - 1) & 1)
    }
    C128_ISSUE_ROW(ys + o + 2, (o + 3) & 3)
    if (o == 0) {
      C128_READ_G(0, 0, 0)
      C128_READ_G(0, 0, 1)
    }
    C128_READ_G((o & 3) * RSTRIDE, 0, 2)
    C128_READ_G((o & 3) * RSTRIDE, 0, 3)

#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][e] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (t == 8 && g == 3) {
          C128_WAIT_G(g, 4);
        } else {
          C128_WAIT_G(g, 6);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          acc[tm] = mfma_h16_32x32x16(wreg[t * 4 + g],
                                                            fa[tm][g], acc[tm]);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < 9) {
          C128_READ_G(((o + (t + 1) / 3) & 3) * RSTRIDE, (t + 1) % 3, g)
        } else if (g < 2) {
          C128_READ_G(((o + 1) & 3) * RSTRIDE, 0, g)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (kh == 0) C128_PUBLISH(acc[1], o & 1)
    else C128_PUBLISH(acc[0], o & 1)
  }
  // last row of the strip
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (kh == 0) C128_EPILOGUE(acc[0], ys + sy - 1, (sy - 1) & 1)
  else C128_EPILOGUE(acc[1], ys + sy - 1, (sy - 1) & 1)
  C64_WAIT_VM(0);
  }  // strips
#undef C128_PUBLISH
#undef C128_EPILOGUE
#undef C128_WAIT_G
#undef C128_READ_G
#undef C128_ISSUE_ROW
}

// ---------------------------------------------------------------------------------------------------------------------
// First VGG layer (Cin = 3, Cout = 64, encoders.py:17-27 / vgg19_bn features[0]) in bf16 mode, straight from the f32 NCHW image:
// replaces im2col (K = 27 -> 32, 64 B / pixel written and read back) + a 256 x 64 GEMM tile pass.  HBM-bound: 12 B / pixel
// in, 128 B / pixel out.  A wave owns 32 consecutive pixels of an image row and walks RY rows down; the 27 taps of a pixel
// are split over the two lanes that share it in the MFMA operand layout (lane (l31, h) supplies k = 16 h + [0, 16) of
// k = ci * 9 + ky * 3 + kx, k >= 27 zero - a permutation of the MFMA's K index applied to both operands, so the weight
// fragment of lane (l31, h), k-step ks is the contiguous 16 bytes W[cout][16 h + 8 ks ...)).  Tap offsets are lane constants
// (two compile-time candidates per slot, selected by h); a row whose 34 x 3 window lies inside the image takes 16 plain
// loads (coalesced over the 32 pixels), any other row the guarded form.  8 packs, 4 MFMAs (2 channel blocks x 2 k-steps),
// bias + ReLU, and the 32 x 128 B of the row go through a per-wave LDS transpose so that every store instruction writes
// whole 128-byte lines (8 lanes per pixel) instead of 32-byte pieces of 32 lines.
__global__ __launch_bounds__(256, 4) void conv3x3_c3_bf16_kernel(const float* __restrict__ img, const bf16_t* __restrict__ w,
                                                                 const float* __restrict__ bias, bf16_t* __restrict__ out, int B,
                                                                 int H, int W, int RY, int nxt, long ntiles) {
  __shared__ __attribute__((aligned(16))) float bias_s[64];
  __shared__ __attribute__((aligned(1024))) unsigned char stage[4][4096];  // per wave: [32 pixels][8 chunks of 16 B], chunk ^ (px & 7)
  if (threadIdx.x < 64) bias_s[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long t = (long)blockIdx.x * 4 + wave;  // (image, row block, 32-pixel column tile)
  if (t >= ntiles) return;
  const int xt = (int)(t % nxt);
  const long r = t / nxt;
  const int nyb = (H + RY - 1) / RY;
  const int y0 = (int)(r % nyb) * RY;
  const int b = (int)(r / nyb);
  const int l31 = lane & 31, h = lane >> 5;
  const int x = xt * 32 + l31;
  const int HW = H * W;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  u32x4 wf[2][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wf[cb][ks] = *reinterpret_cast<const u32x4*>(w + (32 * cb + l31) * 32 + 16 * h + 8 * ks);

  // lane constants: offset of tap s (relative to the pixel, in floats) and the validity bits it needs
  // (bit ky: input row y + ky - 1 inside the image, bit 3 + kx: column x + kx - 1 inside; 64: never)
  int toff[16], tsel[16];
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) {
    const int k0 = s_, k1 = 16 + s_;  // h = 0 / h = 1
    const int o0 = (k0 / 9) * HW + ((k0 % 9) / 3 - 1) * W + (k0 % 3 - 1);
    const int o1 = k1 < 27 ? (k1 / 9) * HW + ((k1 % 9) / 3 - 1) * W + (k1 % 3 - 1) : 0;
    const int m0 = (1 << ((k0 % 9) / 3)) | (8 << (k0 % 3));
    const int m1 = k1 < 27 ? ((1 << ((k1 % 9) / 3)) | (8 << (k1 % 3))) : 64;
    toff[s_] = h ? o1 : o0;
    tsel[s_] = h ? m1 : m0;
  }
  const int colbits = ((x >= 1 && x - 1 < W) ? 8 : 0) | (x < W ? 16 : 0) | (x + 1 < W ? 32 : 0);
  const bool cols_in = xt * 32 >= 1 && xt * 32 + 33 <= W;  // the whole 34-pixel window of the tile (wave-uniform)
  const float* p0 = img + (long)b * 3 * HW;
  bf16_t* outb = out + (long)b * HW * 64;
  const unsigned st0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)stage[wave]);
  const int ny = min(RY, H - y0);
  for (int o = 0; o < ny; ++o) {
    const int y = y0 + o;
    const int base = y * W + x;
    float v[16];
    if (cols_in && y >= 1 && y + 1 < H) {
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) v[s_] = p0[base + toff[s_]];
      if (h) {  // k = 27 .. 31 of the second half do not exist
#pragma unroll
        for (int s_ = 11; s_ < 16; ++s_) v[s_] = 0.f;
      }
    } else {
      const int okbits = colbits | (y >= 1 ? 1 : 0) | 2 | (y + 1 < H ? 4 : 0);
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) {
        const bool ok = (okbits & tsel[s_]) == tsel[s_];
        const float ld = p0[ok ? base + toff[s_] : 0];
        v[s_] = ok ? ld : 0.f;
      }
    }
    u32x4 fr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) fr[ks][j] = pack_bf16x2(v[8 * ks + 2 * j], v[8 * ks + 2 * j + 1]);
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        acc[cb] = mfma_h16_32x32x16(wf[cb][ks],
                                                          fr[ks], acc[cb]);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + 32 * cb + 16 * P + 4 * h);  // channels 32 cb + 8 rg + 4 h + [0, 4)
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_s + 32 * cb + 16 * P + 8 + 4 * h);
        const unsigned a0 = pack_bf16x2(fmaxf(acc[cb][8 * P + 0] + b0[0], 0.f), fmaxf(acc[cb][8 * P + 1] + b0[1], 0.f));
        const unsigned a1 = pack_bf16x2(fmaxf(acc[cb][8 * P + 2] + b0[2], 0.f), fmaxf(acc[cb][8 * P + 3] + b0[3], 0.f));
        const unsigned c0 = pack_bf16x2(fmaxf(acc[cb][8 * P + 4] + b1[0], 0.f), fmaxf(acc[cb][8 * P + 5] + b1[1], 0.f));
        const unsigned c1 = pack_bf16x2(fmaxf(acc[cb][8 * P + 6] + b1[2], 0.f), fmaxf(acc[cb][8 * P + 7] + b1[3], 0.f));
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1, false, false);
        const u32x4 st = {s0[0], s1[0], s0[1], s1[1]};  // channels 32 cb + 16 P + 8 h + [0, 8) of pixel l31: chunk 4 cb + 2 P + h
        const int c = 4 * cb + 2 * P + h;
        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(st0 + l31 * 128 + ((c ^ (l31 & 7)) << 4)) = st;
      }
    // the wave's LDS operations execute in order: the transposed reads below see the writes above
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 st = *reinterpret_cast<__attribute__((address_space(3))) const u32x4*>(st0 + i * 1024 + lane * 16);
      const int px = i * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (px & 7);
      if (xt * 32 + px < W) *reinterpret_cast<u32x4*>(outb + ((long)y * W + xt * 32 + px) * 64 + c * 8) = st;
    }
  }
}

int conv3x3_c3_bf16_launch(const float* img, const void* w, const float* bias, void* out, int B, int H, int W, hipStream_t stream) {
  ROMA_REQUIRE(img && w && bias && out && B > 0 && H > 0 && W > 0, "conv3x3_c3_bf16: bad arguments");
  ROMA_REQUIRE((long)3 * H * W < (1l << 23), "conv3x3_c3_bf16: image too large for the packed tap offsets");
  const int RY = 8;
  const int nxt = (W + 31) / 32;
  const long ntiles = (long)B * ((H + RY - 1) / RY) * nxt;
  ProfScope ps("conv3x3_c3_bf16_kernel", (double)B * H * W * (12.0 + 128.0), "byte", stream);
  hipLaunchKernelGGL(conv3x3_c3_bf16_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, stream, img,
                     reinterpret_cast<const bf16_t*>(w), bias, reinterpret_cast<bf16_t*>(out), B, H, W, RY, nxt, ntiles);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// 0 = launched, 1 = not this kernel's problem
int conv64_try_launch(const GemmArgs& a, hipStream_t stream) {
  static const int use_env = getenv("ROMA_CONV64") ? atoi(getenv("ROMA_CONV64")) : 7;
  const int use = g_conv64_mode >= 0 ? g_conv64_mode : use_env;  // bit 0: Cin = 64 kernels, bit 1: the Cin = 128 kernel
  const bool c128 = a.conv_c == 128;
  if (!(use & (c128 ? 2 : 1))) return 1;
  if ((a.conv_c != 64 && !c128) || a.in_dt != DT_BF16 || a.out_dt != DT_BF16 || a.act != ACT_RELU || !a.bias) return 1;
  if (c128 ? a.N != 128 : (a.N != 64 && a.N != 128)) return 1;
  if (a.ldc != a.N || a.ldw != 9 * a.conv_c || a.batch != 1 || a.mode != EPI_STD || a.scale || a.res || a.res_bf16 || a.alpha != 1.0f)
    return 1;
  const int H = a.conv_h, W = a.conv_w;
  const long hw = (long)H * W;
  if (hw <= 0 || a.M % hw != 0) return 1;
  if ((long)hw * a.conv_c * 2 >= (1l << 31)) return 1;  // 32-bit offsets inside an image's buffer descriptor
  const int B = (int)(a.M / hw);
  const int TW = (!c128 && a.N == 64) ? 128 : 64;
  const int slots = c128 ? 256 : 512;  // persistent workgroups: one (8 waves) or two (4 waves) per CU
  static const int sy_env = getenv("ROMA_CONV64_SY") ? atoi(getenv("ROMA_CONV64_SY")) : 0;
  const int nxt = (W + TW - 1) / TW;
  // strip height: the persistent workgroups take the strips round-robin, so the launch lasts rounds x (SY + ~3 rows of ring
  // prologue); pick the split of H that minimises it (432 rows, 112 columns of strips: 9 strips of 48 rows fill 1.97 rounds
  // of 512, where 14 strips of 32 would leave the fourth round 6 % full)
  int SY = sy_env;
  if (SY <= 0) {
    long best = -1;
    for (int yt = 1; yt <= std::max(1, H / 12); ++yt) {
      const int sy = (H + yt - 1) / yt;
      const long n = (long)B * nxt * ((H + sy - 1) / sy);
      const long cost = ((n + slots - 1) / slots) * (sy + 3);
      if (best < 0 || cost < best) {
        best = cost;
        SY = sy;
      }
    }
  }
  const long nb = (long)B * ((H + SY - 1) / SY) * nxt;
  if (nb <= 0 || nb >= (1l << 30)) return 1;
  char pname[64];
  snprintf(pname, sizeof pname, c128 ? "conv3x3_c128_kernel<%d>" : "conv3x3_c64_kernel<%d>", a.N);
  ProfScope ps(pname, 2.0 * (double)a.M * a.N * 9.0 * a.conv_c, "flop", stream);
  const bf16_t* in = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* w = reinterpret_cast<const bf16_t*>(a.W);
  bf16_t* out = reinterpret_cast<bf16_t*>(a.C);
  constexpr int LDS64 = 4 * 17 * 1024 + 512, LDS64N128 = 4 * 9 * 1024 + 512;  // ring + bias
  constexpr int LDS128 = 4 * 17 * 1024 + 2 * 8 * 4096 + 512;                   // ring + exchange + bias
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS64));
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS64N128));
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c128_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS128));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const long gx = std::min<long>(nb, slots);
  if (c128) {
    hipLaunchKernelGGL(conv3x3_c128_kernel<0>, dim3((unsigned)gx), dim3(512), LDS128, stream, in, w, a.bias, out, B, H, W, SY, nxt, (int)nb);
  } else if (a.N == 64) {
    hipLaunchKernelGGL(conv3x3_c64_kernel<64>, dim3((unsigned)gx), dim3(256), LDS64, stream, in, w, a.bias, out, B, H, W, SY, nxt, (int)nb);
  } else {
    hipLaunchKernelGGL(conv3x3_c64_kernel<128>, dim3((unsigned)gx), dim3(256), LDS64N128, stream, in, w, a.bias, out, B, H, W, SY, nxt, (int)nb);
  }
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
