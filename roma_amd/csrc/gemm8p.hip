// 8-phase bf16 MFMA GEMM for gfx950: the large, MFMA-bound problems of the match() path (DINOv2 / decoder linears,
// stride-16 refiner 1x1, VGG 3x3 implicit GEMM with >= 256 output channels).  C[M,N] = epi(A[M,K] . W[N,K]^T).
//
// Why a second main loop (gemm.hip keeps every other shape): gemm.hip's loop has ONE barrier per 64-deep K slab, both
// wave groups in lock-step, a vmcnt(0) drain per slab and ~1000 instructions of branchy per-slab DMA address selection
// (tails, zero page, conv taps).  SQ counters (profiles/r01_pmc_sq_summary.json): 39 % of the wave cycles parked at
// s_waitcnt / s_barrier, 33 % issue stalls, MFMA busy 0.33.  This kernel follows the "256^2 8-phase" anatomy of
// /opt/skills/guides/cdna_hip_programming.md section 5, restated for the 32x32x16 MFMA and 128-byte swizzled LDS rows:
//
//   * 256 x 256 x 64 tiles, 8 waves = 2 wave groups (wr) x 4 column slices (wc); waves w and w + 4 share a SIMD and
//     belong to different groups.  Group 1 runs ONE s_barrier behind group 0, so between two consecutive barriers one
//     group is in its load block (LDS fragment reads + one half-tile of LDS-DMA) while the other is in its MFMA block:
//     the matrix pipe of every SIMD always has a wave with operands in registers.
//   * wave (wr, wc) owns the contiguous 128 x 64 block rows [128 wr, +128) x cols [64 wc, +64) of the tile (so the
//     staged row-writer epilogues of gemm_device.h apply unchanged), as four 64 x 32 quadrants (mh, nh).  The LDS image
//     is permuted so that quadrant operands are whole half-tiles: A half mh = rows {128 wr + 64 mh + [0, 64)},
//     W half nh = rows {64 wc + 32 nh + [0, 32)}.  LDS-DMA writes lane-linear, the SOURCE address is per lane, so the
//     permutation (and the XOR bank swizzle) costs nothing.
//   * one K tile = 4 phases: P1 reads A0 + W0 -> quadrant (0,0); P2 reads W1 -> (0,1); P3 reads A1 -> (1,1);
//     P4 reads nothing -> (1,0).  Every phase: [fragment reads][DMA of one half-tile][vmcnt at P4] s_barrier,
//     lgkmcnt(0), 8 MFMAs, s_barrier.
//   * LDS-DMA runs 1.5 K tiles ahead of the math through two 64 KB buffers, as ONE stream over all the (tile, k)
//     positions a persistent workgroup will visit - P1(s): W1(s+1), P2(s): A1(s+1), P3(s): A0(s+2), P4(s): W0(s+2),
//     then vmcnt(4) - so the first K tiles of the next output tile arrive under the current tile's last MFMAs and its
//     epilogue.  A region is re-staged >= 2 phases after its last read; the counted vmcnt at P4 leaves only the two
//     half-tiles of s+2 in flight and sits before P4's first barrier, the first read of s+1 is in the next phase
//     (wait -> barrier -> read).  vmcnt also counts the epilogue's global stores; they are older than the DMA pieces the
//     wait must leave in flight, so a counted wait stays conservative.
//   * the per-phase DMA issue is 2 instructions + 2 64-bit adds per wave: M / N tails and padded QKV tokens point at
//     a zero buffer (fixed once per tile), K must be a multiple of 64; the 3x3 convolution selects {pixel + tap offset,
//     zero page} per piece from a 9-bit validity mask computed once per tile.
//   * fragment reads are inline asm (hipcc would drain the DMA queue before any LDS read it can see).  The wait that
//     covers them names every destination register as a read-write operand, so no consumer can be scheduled above it;
//     tools/audit_gemm8p_isa.py checks in the emitted ISA that nothing touches those registers between a read and its
//     wait.
// packed-f32 GELU epilogue (gemm_device.h): this translation unit is compiled with packed-f32 instructions enabled
// (Makefile PK_SRCS) and is covered by tools/audit_pk_sgpr.py
#define ROMA_EPI_PK 1
// Round 6: the dense operands go through buffer descriptors - buffer_load_dwordx4 ... lds with an SGPR base, a 32-bit lane
// offset fixed per tile and the K offset in an SGPR - instead of flat 64-bit lane addresses rebuilt per piece: no VALU in the
// DMA issue, and rows outside M / N (and padded QKV tokens) carry an offset beyond num_records, which the hardware answers
// with zeros.  fc2 250 -> 229 us, qkv 204 -> 188, proj 68 -> 63, step -1.35 ms, bit-identical (profiles/r06_v23_*).  The 3x3
// convolution form (tap offsets per piece) keeps the flat addresses.  Undefine for the A/B.
#define ROMA_R8_BUFLDS 1
#include "gemm.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "conv_patch.h"
#include "gemm_device.h"

namespace roma {

// zero source for out-of-range rows of the dense form: the per-k-tile byte offset (k * 128) is added to EVERY lane's
// pointer, so the zero "row" must be as long as the longest K row (K <= 32704 bf16)
static __device__ __attribute__((aligned(256))) unsigned int g_zero_rows[16384];

int g_gemm8p_maxwg = -1;  // roma_tuning("gemm8p_maxwg"): see launch8p_s
int g_gemm_tuning[2] = {-1, -1};  // [0] gemm8p on / off, [1] dbg bits; -1 = environment (roma_tuning, tests / A-B runs)

enum { E8_NONE = 0, E8_RELU = 1, E8_GELU = 2, E8_RESBF16 = 3, E8_QKV = 4 };

#define R8_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define R8_DS_READ(REG, ADDR, OFF)                                                               \
  if constexpr (x_rd) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(REG) : "v"(ADDR), "n"(OFF)); \
  else asm volatile("" : "=v"(REG)); /* no-read ablation: the register is (re)defined here, with whatever it holds */

// ablation / trace build (ABL, tools/bench_gemm_ablation.py): phase time stamps of waves 0 and 4 of the first 16 workgroups
// [workgroup][wave group][K tile < 256][phase] -> low 32 bits of s_memtime at the phase's first barrier release
__device__ unsigned int g_gemm_trace[16 * 2 * 256 * 4];

// SCHED 0: quadrant phases (reads 12 / 4 / 8 / 0 per phase).  SCHED 1 ("KH"): k-half phases (reads 8 / 6 / 6 / 4), see
// the loop body.  ABL (measuring tool only, compile-time so that the loop carries no branches): 1 = phase trace above,
// 2 = no fragment reads, 4 = no LDS-DMA, 8 = no MFMA (selected by a.dbg bits 32768 + 4096 / 8192 / 16384).
template <typename TOUT, bool CONV, int EPI, bool DMAMF, int SCHED = 0, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const GemmArgs a) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int TILE_A = BM * ROWB, BUF = TILE_A + BN * ROWB;  // 64 KiB per K tile
  constexpr int TM = 4, TN = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- persistent tile walk, XCD-aware (workgroup b runs on XCD b % 8; each XCD walks a contiguous band, n fastest)
  const int NT = (a.N + BN - 1) / BN;
  const long nblk = (long)((a.M + BM - 1) / BM) * NT;
  const long per_xcd = (nblk + 7) / 8;
  const int xcd = blockIdx.x % 8;
  const long wg_per_xcd = gridDim.x / 8;
  long li = blockIdx.x / 8;
  if (li >= per_xcd || (long)xcd * per_xcd + li >= nblk) return;

  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* Wb = reinterpret_cast<const bf16_t*>(a.W);
  const char* zrows = reinterpret_cast<const char*>(g_zero_rows);
  const int nk = (a.dbg & 2) ? 2 : a.K / BK;  // dbg 2: two K tiles only (isolates the per-tile overhead in A/B runs)

  // ---- LDS-DMA descriptors.  A half-tile is 128 LDS rows = 16 pieces of 8 rows; wave w stages pieces 2w, 2w + 1 of
  // every half-tile.  lane -> (row r8 of the piece, 16-byte slot); the slot holds source chunk slot ^ ((row >> 1) & 7).
  const char* a_src[2][2];  // [half][piece]
  const char* w_src[2][2];
  unsigned a_mask[2][2];    // conv: bit t = tap t of this row is inside the image
#ifdef ROMA_R8_BUFLDS
  // dense operands through buffer descriptors (see the top of the file): byte offsets of this lane's pieces, 0x80000000 = outside
  unsigned a_off[2][2], w_off[2][2];
  const long a_rows_ = a.qkv_pad ? (long)(a.M / a.npad) * a.ntok : (long)a.M;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, CONV ? 0 : (int)(((a_rows_ - 1) * a.lda + a.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)((((long)a.N - 1) * a.ldw + a.K) * 2), 0x00020000);
#define R8_BL16(RS, VOFF, SOFF, DST) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(DST), 16, (int)(VOFF), (int)(SOFF), 0, 0)
#endif
  // descriptors of tile (TMI, TNI) - the DMA stream runs ahead of the math, so they may describe the NEXT output tile
#ifdef ROMA_R8_BUFLDS
#define R8_SETUP_W(hf, j, gn, chunk) \
  w_off[hf][j] = gn < a.N ? (unsigned)(((long)gn * a.ldw + chunk * 8) * 2) : 0x80000000u; \
  w_src[hf][j] = zrows;
#define R8_SETUP_A_DENSE(hf, j, gm, chunk)                                                                    \
  {                                                                                                           \
    unsigned o_ = 0x80000000u;                                                                                \
    if (gm < a.M) {                                                                                           \
      if (a.qkv_pad) {                                                                                        \
        const int qb_ = gm / a.npad;                                                                          \
        const int qt_ = gm - qb_ * a.npad;                                                                    \
        if (qt_ < a.ntok) o_ = (unsigned)((((long)qb_ * a.ntok + qt_) * a.lda + chunk * 8) * 2);              \
      } else {                                                                                                \
        o_ = (unsigned)(((long)gm * a.lda + chunk * 8) * 2);                                                  \
      }                                                                                                       \
    }                                                                                                         \
    a_off[hf][j] = o_;                                                                                        \
    a_src[hf][j] = zrows;                                                                                     \
  }
#else
#define R8_SETUP_W(hf, j, gn, chunk) \
  w_src[hf][j] = gn < a.N ? reinterpret_cast<const char*>(Wb + (long)gn * a.ldw + chunk * 8) : zrows + chunk * 16;
#define R8_SETUP_A_DENSE(hf, j, gm, chunk)                                                                    \
  {                                                                                                           \
    const char* p = zrows + chunk * 16;                                                                       \
    if (gm < a.M) {                                                                                           \
      if (a.qkv_pad) { /* rows = (image, padded token); tokens >= ntok read zeros */                          \
        const int qb_ = gm / a.npad;                                                                          \
        const int qt_ = gm - qb_ * a.npad;                                                                    \
        if (qt_ < a.ntok) p = reinterpret_cast<const char*>(Ab + ((long)qb_ * a.ntok + qt_) * a.lda + chunk * 8); \
      } else {                                                                                                \
        p = reinterpret_cast<const char*>(Ab + (long)gm * a.lda + chunk * 8);                                 \
      }                                                                                                       \
    }                                                                                                         \
    a_src[hf][j] = p;                                                                                         \
  }
#endif
#define R8_TILE_SETUP(TMI, TNI)                                                                               \
  {                                                                                                           \
    const int d_m0 = (TMI) * BM, d_n0 = (TNI) * BN;                                                           \
    /* opaque lane id: everything derived from it is computed HERE, once per tile, instead of being hoisted to */ \
    /* kernel entry and kept (or spilled) across the K loop, which has no registers to spare                    */ \
    int ln_ = lane;                                                                                           \
    asm volatile("" : "+v"(ln_));                                                                             \
    const int r8 = ln_ >> 3, slot = ln_ & 7;                                                                  \
    _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) _Pragma("unroll") for (int j = 0; j < 2; ++j) {          \
      const int i = 16 * wave + 8 * j + r8; /* LDS row inside the half-tile */                                \
      const int chunk = slot ^ ((i >> 1) & 7);                                                                \
      const int gm = d_m0 + (i >> 6) * 128 + hf * 64 + (i & 63); /* M < 2^31: 32-bit divisions below */        \
      const int gn = d_n0 + (i >> 5) * 64 + hf * 32 + (i & 31);                                               \
      R8_SETUP_W(hf, j, gn, chunk)                                                                            \
      if constexpr (CONV) {                                                                                   \
        unsigned mk = 0;                                                                                      \
        const char* p = zrows;                                                                                \
        if (gm < a.M) {                                                                                       \
          const int hw = a.conv_h * a.conv_w;                                                                 \
          const int b = gm / hw;                                                                              \
          const int rem = gm - b * hw;                                                                        \
          const int y = rem / a.conv_w, x = rem - y * a.conv_w;                                               \
          _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                     \
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;                                                 \
            if (yy >= 0 && yy < a.conv_h && xx >= 0 && xx < a.conv_w) mk |= 1u << t;                          \
          }                                                                                                   \
          p = reinterpret_cast<const char*>(Ab + (long)gm * a.conv_c + chunk * 8);                            \
        }                                                                                                     \
        a_mask[hf][j] = mk;                                                                                   \
        a_src[hf][j] = p;                                                                                     \
      } else {                                                                                                \
        a_mask[hf][j] = 0;                                                                                    \
        R8_SETUP_A_DENSE(hf, j, gm, chunk)                                                                    \
      }                                                                                                       \
    }                                                                                                         \
  }

#ifdef ROMA_R8_BUFLDS
#define R8_ISSUE_DENSE(RS, OFF, SRC, HF, KP, DST) \
  { const int so_ = (KP) * (BK * 2); R8_BL16(RS, OFF[HF][0], so_, DST); R8_BL16(RS, OFF[HF][1], so_, (DST) + 1024); }
#else
#define R8_ISSUE_DENSE(RS, OFF, SRC, HF, KP, DST) \
  { const long soff_ = (long)(KP) * (BK * 2); glds16(SRC[HF][0] + soff_, DST); glds16(SRC[HF][1] + soff_, (DST) + 1024); }
#endif
  // K position KP (in K tiles, of the DMA tile) of A half HF into LDS buffer BSEL
#define R8_ISSUE_A(HF, KP, BSEL)                                                                              \
  {                                                                                                           \
    char* dst_ = smem + (BSEL) * BUF + (HF) * 128 * ROWB + (2 * wave) * 1024;                                 \
    if constexpr (CONV) {                                                                                     \
      const int slab_ = ((KP) * 7282) >> 16; /* KP / 9 (exact for KP < 1024; KP < 72 here) */                 \
      const int tap_ = a.conv_korder ? (KP) - 9 * slab_ : ((KP) * conv_inv) >> 16;                            \
      const int c0_ = (a.conv_korder ? slab_ : (KP) - tap_ * conv_spt) * BK;                                  \
      const int dy_ = tap_ / 3 - 1, dx_ = tap_ - (tap_ / 3) * 3 - 1;                                          \
      const long soff_ = (((long)dy_ * a.conv_w + dx_) * a.conv_c + c0_) * 2;                                 \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                         \
        const bool ok_ = (a_mask[HF][j] >> tap_) & 1u;                                                        \
        glds16(ok_ ? a_src[HF][j] + soff_ : zrows + (lane & 7) * 16, dst_ + j * 1024);                        \
      }                                                                                                       \
    } else {                                                                                                  \
      R8_ISSUE_DENSE(rs_a, a_off, a_src, HF, KP, dst_)                                                        \
    }                                                                                                         \
  }
#define R8_ISSUE_W(HF, KP, BSEL)                                                                              \
  {                                                                                                           \
    char* dst_ = smem + (BSEL) * BUF + TILE_A + (HF) * 128 * ROWB + (2 * wave) * 1024;                        \
    R8_ISSUE_DENSE(rs_w, w_off, w_src, HF, KP, dst_)                                                          \
  }
  // one piece (J = 0, 1) of a half-tile: the DMAMF schedule places the two pieces between the MFMAs of a phase
#define R8_ISSUE_A1(HF, J, KP, BSEL)                                                                          \
  {                                                                                                           \
    char* dst_ = smem + (BSEL) * BUF + (HF) * 128 * ROWB + (2 * wave + (J)) * 1024;                           \
    if constexpr (CONV) {                                                                                     \
      const int slab_ = ((KP) * 7282) >> 16; /* KP / 9 (exact for KP < 1024; KP < 72 here) */                 \
      const int tap_ = a.conv_korder ? (KP) - 9 * slab_ : ((KP) * conv_inv) >> 16;                            \
      const int c0_ = (a.conv_korder ? slab_ : (KP) - tap_ * conv_spt) * BK;                                  \
      const int dy_ = tap_ / 3 - 1, dx_ = tap_ - (tap_ / 3) * 3 - 1;                                          \
      const long soff_ = (((long)dy_ * a.conv_w + dx_) * a.conv_c + c0_) * 2;                                 \
      const bool ok_ = (a_mask[HF][J] >> tap_) & 1u;                                                          \
      glds16(ok_ ? a_src[HF][J] + soff_ : zrows + (lane & 7) * 16, dst_);                                     \
    } else {                                                                                                  \
      glds16(a_src[HF][J] + (long)(KP) * (BK * 2), dst_);                                                     \
    }                                                                                                         \
  }
#define R8_ISSUE_W1(HF, J, KP, BSEL) \
  glds16(w_src[HF][J] + (long)(KP) * (BK * 2), smem + (BSEL) * BUF + TILE_A + (HF) * 128 * ROWB + (2 * wave + (J)) * 1024);
  const int conv_spt = CONV ? a.conv_c / BK : 1;                  // K tiles per 3x3 tap
  const int conv_inv = CONV ? 65536 / conv_spt + 1 : 0;           // tap = (k * inv) >> 16, exact for k < 9 * spt <= 72

  // ---- fragment read addresses: row = block_row0 + l31 (block_row0 % 32 == 0), 16-byte slot (2g + h) ^ ((l31 >> 1) & 7)
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const int sw = (l31 >> 1) & 7;
  unsigned rd[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) rd[g] = (unsigned)(l31 * ROWB + (((2 * g + h) ^ sw) << 4));
  const unsigned a_row0 = (unsigned)((64 * wr) * ROWB);           // + mh * 128 rows + mt * 32 rows
  const unsigned w_row0 = (unsigned)(TILE_A + (32 * wc) * ROWB);  // + nh * 128 rows

  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 af[2][4], bf0[4], bf1[4];  // A fragments [mt][k-group] of the current 64-row half; W fragments of both halves
  // SCHED 1: A fragments [mt][k-group of the pair] of the current (half, k-pair); W halves 0 / 1 of k-pair 0 (fw*) and 1 (fw*n)
  u32x4 fa[2][2], fw0[2], fw1[2], fw0n[2], fw1n[2];
  constexpr bool x_rd = !(ABL & 2), x_dma = !(ABL & 4), x_mf = !(ABL & 8);
  unsigned long long tstamp[4] = {0, 0, 0, 0};

#define R8_READ_A(MH, SB)                                                                                     \
  _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int g = 0; g < 4; ++g)               \
      R8_DS_READ(af[mt][g], (SB) + a_row0 + rd[g], ((MH) * 128 + mt * 32) * ROWB);
#define R8_READ_W(BF, NH, SB) \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) R8_DS_READ(BF[g], (SB) + w_row0 + rd[g], ((NH) * 128) * ROWB);
  // waits: every register the covered reads write is a read-write operand, so no consumer can move above the wait
#define R8_WAIT_LGKM_W(BF) \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(BF[0]), "+v"(BF[1]), "+v"(BF[2]), "+v"(BF[3])::"memory")
#define R8_WAIT_LGKM_A()                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
               : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]), "+v"(af[1][1]), \
                 "+v"(af[1][2]), "+v"(af[1][3])::"memory")
#define R8_WAIT_LGKM_AW(BF)                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
               : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]), "+v"(af[1][1]), \
                 "+v"(af[1][2]), "+v"(af[1][3]), "+v"(BF[0]), "+v"(BF[1]), "+v"(BF[2]), "+v"(BF[3])::"memory")
  // D[n][m] += W[n][k] A[m][k]: the W fragment is the first operand, so a lane owns 4 consecutive n of one m
#define R8_MFMA_Q(MH, NH, BF)                                                                                  \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                 \
      acc[NH][(MH) * 2 + mt] = mfma_h16_32x32x16(BF[g],    \
                                                                       af[mt][g], \
                                                                       acc[NH][(MH) * 2 + mt]);
#define R8_MFMA_G(MH, NH, BF, G)                                                                               \
  _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                              \
      acc[NH][(MH) * 2 + mt] = mfma_h16_32x32x16(BF[G],    \
                                                                       af[mt][G], \
                                                                       acc[NH][(MH) * 2 + mt]);
  // ---- SCHED 1 ("KH") fragment reads / waits / MFMA blocks
#define R8K_READ_A(MH, GP, SB)                                                                                \
  _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int gl = 0; gl < 2; ++gl)            \
      R8_DS_READ(fa[mt][gl], (SB) + a_row0 + rd[2 * (GP) + gl], ((MH) * 128 + mt * 32) * ROWB);
#define R8K_READ_W(DST, NH, GP, SB) \
  _Pragma("unroll") for (int gl = 0; gl < 2; ++gl) R8_DS_READ(DST[gl], (SB) + w_row0 + rd[2 * (GP) + gl], ((NH) * 128) * ROWB);
#define R8K_WAIT_A() \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1])::"memory")
#define R8K_WAIT_AW(W)                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(W[0]), "+v"(W[1])::"memory")
#define R8K_WAIT_AWW(W, V)                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(W[0]), "+v"(W[1]), "+v"(V[0]), \
                 "+v"(V[1])::"memory")
  // 8 MFMAs over the FOUR accumulators of row half MH (each reused after 4 MFMAs); k order per accumulator unchanged
#define R8K_MFMA(MH, W0, W1)                                                                                   \
  if (x_mf) {                                                                                                  \
    _Pragma("unroll") for (int gl = 0; gl < 2; ++gl) {                                                         \
      _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                         \
          acc[0][(MH) * 2 + mt] = mfma_h16_32x32x16(W0[gl], fa[mt][gl], acc[0][(MH) * 2 + mt]);               \
      _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                         \
          acc[1][(MH) * 2 + mt] = mfma_h16_32x32x16(W1[gl], fa[mt][gl], acc[1][(MH) * 2 + mt]);               \
    }                                                                                                          \
  }
  // DMAMF phase (experiment, off by default): the two LDS-DMA pieces of the phase are issued BETWEEN the MFMAs (after the
  // 2nd and the 4th of 8) instead of in the load block.  The idea: an LDS-DMA instruction costs the issuing wave ~100-180
  // cycles in a block that also carries the fragment reads but ~60 among bare MFMAs (MI355X_MICROARCH.md).  Measured: no
  // gain at K = 1024 and a loss at long K / on the convolution - the matrix pipe is better off with an uninterrupted
  // 8-MFMA burst than with a shorter partner load block.
#define R8_PHASE_DM(WAIT, MH, NH, BF, ISS0, ISS1) \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  WAIT;                                          \
  __builtin_amdgcn_sched_barrier(0);             \
  if (prio) __builtin_amdgcn_s_setprio(1);       \
  R8_MFMA_G(MH, NH, BF, 0)                       \
  __builtin_amdgcn_sched_barrier(0);             \
  ISS0                                           \
  __builtin_amdgcn_sched_barrier(0);             \
  R8_MFMA_G(MH, NH, BF, 1)                       \
  __builtin_amdgcn_sched_barrier(0);             \
  ISS1                                           \
  __builtin_amdgcn_sched_barrier(0);             \
  R8_MFMA_G(MH, NH, BF, 2)                       \
  R8_MFMA_G(MH, NH, BF, 3)                       \
  if (prio) __builtin_amdgcn_s_setprio(0);       \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  __builtin_amdgcn_sched_barrier(0);
#define R8_PHASE(WAIT, MF) R8_PHASE_T(WAIT, MF, 0)
#define R8_PHASE_T(WAIT, MF, PH)                 \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  if constexpr ((ABL & 1) != 0) tstamp[PH] = __builtin_readcyclecounter(); \
  WAIT;                                          \
  __builtin_amdgcn_sched_barrier(0);             \
  if (prio) __builtin_amdgcn_s_setprio(1);       \
  MF;                                            \
  if (prio) __builtin_amdgcn_s_setprio(0);       \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  __builtin_amdgcn_sched_barrier(0);

#ifdef ROMA_TOOLS_BUILD  // A/B switches of tools builds: no s_setprio around the MFMA blocks / no wave-group stagger
  const bool prio = !(a.dbg & 64);
  const bool stagger = !(a.dbg & 128);
#else  // shipped libraries: constants, so that the K loop carries no branch around its s_setprio pairs
  constexpr bool prio = true, stagger = true;
#endif

  // walk order inside the XCD's band: groups of `gm` tile rows, m fastest inside a group - the 32 workgroups of an XCD that run
  // side by side then cover a gm x (32 / gm) block of tiles ((gm + 32 / gm) operand panels per K step through the XCD's L2)
  // instead of 32 / NT rows x NT columns.  gm = 1 is the plain row-major walk.  One division pair per output tile.
  const int MT = (a.M + BM - 1) / BM;
  const int gm = a.walk_gm > 1 ? a.walk_gm : 1;
#define R8_TILE_OF(L, TMO, TNO)                                       \
  {                                                                   \
    const long l_ = (L);                                              \
    const int g_ = (int)(l_ / ((long)gm * NT));                       \
    const int r_ = (int)(l_ - (long)g_ * gm * NT);                    \
    const int ge_ = min(gm, MT - g_ * gm);                            \
    TNO = r_ / ge_;                                                   \
    TMO = g_ * gm + (r_ - TNO * ge_);                                 \
  }
  int c_tm, c_tn;
  R8_TILE_OF((long)xcd * per_xcd + li, c_tm, c_tn)

  // ---- prologue: K tile 0 complete, A0 / W0 of K tile 1 under way (nk >= 2 is guaranteed by the dispatcher)
  R8_TILE_SETUP(c_tm, c_tn)
  if constexpr (SCHED == 1) {
    // KH stream order: ... A0(s) W0(s) W1(s) A1(s) | W0(s+1): K tile 0 complete, W0 of K tile 1 under way; the wait leaves
    // A1(0) and W0(1) in flight (A1(0) is covered by the P1 wait of the first K tile)
    R8_ISSUE_A(0, 0, 0) R8_ISSUE_W(0, 0, 0) R8_ISSUE_W(1, 0, 0) R8_ISSUE_A(1, 0, 0)
    R8_ISSUE_W(0, 1, 1)
  } else {
    R8_ISSUE_A(0, 0, 0) R8_ISSUE_W(0, 0, 0) R8_ISSUE_W(1, 0, 0) R8_ISSUE_A(1, 0, 0)
    R8_ISSUE_A(0, 1, 1) R8_ISSUE_W(0, 1, 1)
  }
  R8_WAIT_VM(4);
  __builtin_amdgcn_s_barrier();
  if (stagger && wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0 from here on

  unsigned gk = 0;  // stream position of the math (parity = LDS buffer)
  for (;;) {
    const long m0 = (long)c_tm * BM;
    const int n0 = c_tn * BN;
    const long li_next = li + wg_per_xcd;
    const bool has_next = li_next < per_xcd && (long)xcd * per_xcd + li_next < nblk;
    int n_tm = 0, n_tn = 0;
    if (has_next) R8_TILE_OF((long)xcd * per_xcd + li_next, n_tm, n_tn)

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // The DMA descriptors always describe the math's own tile at kt == 0 (the stream moved on to it at kt == nk - 2 of
    // the previous tile).  Rebuilding them here - a few dozen VALU - instead of carrying them keeps their 20 registers
    // dead across the epilogue, which needs every register it can get (128 accumulators + the per-column vectors).
    if (gk != 0) R8_TILE_SETUP(c_tm, c_tn)

    // Round 6: K tiles with kt + 2 < nk run the STEADY copy of the body (constant stream bookkeeping), the last two of
    // every output tile the general one - gemm8p_ktile.inc
    int kt = 0;
    for (; kt + 2 < nk; ++kt, ++gk) {
#define R8_STEADY 1
#include "gemm8p_ktile.inc"
#undef R8_STEADY
    }
    for (; kt < nk; ++kt, ++gk) {
#define R8_STEADY 0
#include "gemm8p_ktile.inc"
#undef R8_STEADY
    }
    li = li_next;
    c_tm = n_tm;
    c_tn = n_tn;

    // ---------------------------------------------------------------- epilogue (staged row writers, gemm_device.h)
    if (!(a.dbg & 256)) {  // dbg 256: no epilogue at all (tuning experiments)
      constexpr int SLICE = 4096;  // bf16: 32 rows x 128 B of the wave's 64 columns; f32: one 32 x 32 block
      char* ws = smem + 2 * BUF + wave * SLICE;
      const long mw0 = m0 + (long)wr * 128;
      const int nw0 = n0 + wc * 64;
      int lane_e = lane;  // opaque copy: the epilogue's lane-dependent addresses are built here, not before the K loop
      asm volatile("" : "+v"(lane_e));
      const bool full_tile = m0 + BM <= a.M && n0 + BN <= a.N;
      if constexpr (sizeof(TOUT) == 2) {
        bf16_t* Cbb = reinterpret_cast<bf16_t*>(a.C);
        if constexpr (EPI == E8_QKV) {
          epi_staged_qkv<TM, TN>(acc, a, ws, mw0, nw0, lane_e);
        } else if constexpr (EPI == E8_GELU) {
          if (full_tile) epi_staged_bf16<TM, TN, ACT_GELU, true>(acc, a, Cbb, ws, mw0, nw0, lane_e);
          else epi_staged_bf16<TM, TN, ACT_GELU, false>(acc, a, Cbb, ws, mw0, nw0, lane_e);
        } else if constexpr (EPI == E8_RELU) {
          if (full_tile) epi_staged_bf16<TM, TN, ACT_RELU, true>(acc, a, Cbb, ws, mw0, nw0, lane_e);
          else epi_staged_bf16<TM, TN, ACT_RELU, false>(acc, a, Cbb, ws, mw0, nw0, lane_e);
        } else if constexpr (EPI == E8_RESBF16) {
          const bf16_t* Rbb = reinterpret_cast<const bf16_t*>(a.res_bf16);
          if (full_tile) epi_staged_bf16<TM, TN, ACT_NONE, true, true>(acc, a, Cbb, ws, mw0, nw0, lane_e, Rbb);
          else epi_staged_bf16<TM, TN, ACT_NONE, false, true>(acc, a, Cbb, ws, mw0, nw0, lane_e, Rbb);
        } else {
          if (full_tile) epi_staged_bf16<TM, TN, ACT_NONE, true>(acc, a, Cbb, ws, mw0, nw0, lane_e);
          else epi_staged_bf16<TM, TN, ACT_NONE, false>(acc, a, Cbb, ws, mw0, nw0, lane_e);
        }
      } else {
        float* Cbf = reinterpret_cast<float*>(a.C);
        if constexpr (EPI == E8_RELU) {
          if (full_tile) epi_staged_f32<TM, TN, ACT_RELU, true>(acc, a, Cbf, a.res, ws, mw0, nw0, lane_e);
          else epi_staged_f32<TM, TN, ACT_RELU, false>(acc, a, Cbf, a.res, ws, mw0, nw0, lane_e);
        } else {
          if (full_tile) epi_staged_f32<TM, TN, ACT_NONE, true>(acc, a, Cbf, a.res, ws, mw0, nw0, lane_e);
          else epi_staged_f32<TM, TN, ACT_NONE, false>(acc, a, Cbf, a.res, ws, mw0, nw0, lane_e);
        }
      }
    }
    if constexpr (sizeof(TOUT) == 4) {
      // The f32 row writer's partial-tile variant spills a few address registers; hipcc then carries "scratch reload
      // pending" into the K loop and puts an s_waitcnt vmcnt(0) at its top (every K tile: the LDS-DMA pipeline drained).
      // One explicit vmcnt(0) per TILE, in a form the compiler sees, clears that bookkeeping; it costs the drain of this
      // tile's stores (the first P4 wait of the next tile would sit through most of it anyway).
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    if (!has_next) break;
  }
  if (stagger && wr == 0) __builtin_amdgcn_s_barrier();  // group 0 meets group 1's extra barrier
  if constexpr ((ABL & 1) != 0) {
    __syncthreads();
    if (blockIdx.x < 16 && (wave == 0 || wave == 4)) {
      const unsigned* src = reinterpret_cast<const unsigned*>(smem + 2 * BUF + wave * 4096);
      unsigned* dst = g_gemm_trace + ((blockIdx.x * 2 + (wave >> 2)) * 256) * 4;
      for (int i = lane; i < 1024; i += 64) dst[i] = src[i];
    }
  }
#undef R8_TILE_OF
#undef R8_PHASE_T
#undef R8K_MFMA
#undef R8K_WAIT_AWW
#undef R8K_WAIT_AW
#undef R8K_WAIT_A
#undef R8K_READ_W
#undef R8K_READ_A
#undef R8_PHASE
#undef R8_PHASE_DM
#undef R8_MFMA_G
#undef R8_ISSUE_W1
#undef R8_ISSUE_A1
#undef R8_MFMA_Q
#undef R8_WAIT_LGKM_AW
#undef R8_WAIT_LGKM_A
#undef R8_WAIT_LGKM_W
#undef R8_READ_W
#undef R8_READ_A
#undef R8_ISSUE_W
#undef R8_ISSUE_A
#undef R8_TILE_SETUP
#undef R8_ISSUE_DENSE
#undef R8_SETUP_A_DENSE
#undef R8_SETUP_W
}

template <typename TOUT, bool CONV, int EPI, bool DMAMF = false, int SCHED = 0, int ABL = 0>
static int launch8p_s(const GemmArgs& a, hipStream_t stream, const char* epi_name) {
  constexpr int BM = 256, BN = 256;
  const long nblk = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const size_t lds = (size_t)2 * (BM + BN) * ROWB + 8 * 4096;  // 160 KiB: one persistent workgroup per CU
  // roma_tuning("gemm8p_maxwg", n): tools only - cap the persistent grid (a multiple of 8) to measure what a tile's epilogue
  // costs when fewer workgroups store at the same time (tools/bench_gemm_burst.py)
  const long cap = g_gemm8p_maxwg >= 8 ? (g_gemm8p_maxwg / 8) * 8 : 256;
  const long gx = std::min<long>(((nblk + 7) / 8) * 8, std::min<long>(cap, 256));
  char pname[96];
  snprintf(pname, sizeof pname, "gemm8p_kernel<" ROMA_H16_NAME ",%s,%s,%s>", sizeof(TOUT) == 4 ? "f32" : ROMA_H16_NAME, CONV ? "conv3x3" : "dense", epi_name);
  ProfScope ps(pname, 2.0 * (double)(a.m_alg > 0 ? a.m_alg : a.M) * (a.n_alg > 0 ? a.n_alg : a.N) * (a.k_alg > 0 ? a.k_alg : a.K), "flop", stream);
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel<TOUT, CONV, EPI, DMAMF, SCHED, ABL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm8p_kernel<TOUT, CONV, EPI, DMAMF, SCHED, ABL>), dim3((unsigned)gx), dim3(512), lds, stream, a);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// schedule selection: roma_tuning("gemm8p_sched", v) (-1 = environment ROMA_GEMM8P_SCHED, default 1 = k-half phases; 0 =
// quadrant phases); dbg bit 65536 flips it for one launch (A/B inside one process)
int g_gemm8p_sched = -1;
static int gemm8p_sched_of(const GemmArgs& a) {
  static const int sched_env = getenv("ROMA_GEMM8P_SCHED") ? atoi(getenv("ROMA_GEMM8P_SCHED")) : 1;
  const int base = g_gemm8p_sched >= 0 ? g_gemm8p_sched : sched_env;
  return (base ? 1 : 0) ^ ((a.dbg & 65536) ? 1 : 0);
}
template <typename TOUT, bool CONV, int EPI>
static int launch8p(const GemmArgs& a, hipStream_t stream, const char* epi_name) {
#ifdef ROMA_TOOLS_BUILD  // the quadrant-phase schedule of rounds 2-3 (bit-identical; A/B only) is not in the shipped libraries
  if (!gemm8p_sched_of(a)) return launch8p_s<TOUT, CONV, EPI, false, 0>(a, stream, epi_name);
#else
  ROMA_REQUIRE(gemm8p_sched_of(a) == 1, "gemm8p: the quadrant-phase schedule (gemm8p_sched = 0) is an A/B variant of tools builds (make TOOLS=1)");
#endif
  return launch8p_s<TOUT, CONV, EPI, false, 1>(a, stream, epi_name);
}

int gemm8p_trace_read(unsigned* host, long n) {
  if (n > (long)(sizeof(unsigned) * 16 * 2 * 256 * 4)) n = (long)(sizeof(unsigned) * 16 * 2 * 256 * 4);
  ROMA_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_trace), (size_t)n));
  return 0;
}

// Decides whether this problem belongs to the 8-phase kernel and launches it; returns 1 when it did not take the
// problem (the caller falls back to gemm.hip), 0 on success, < 0 on error.  `a` is the argument block AFTER
// gemm_launch's own normalisation (qkv_pad / m_alg already applied).
int conv64_try_launch(const GemmArgs& a, hipStream_t stream);  // conv64.hip (weight-stationary 3x3, Cin = 64)

int g_gemm8p_walk = -1;  // roma_tuning("gemm8p_walk", n): tile rows per walk group (1 = row major); -1 = the dispatcher's choice
int gemm8p_try_launch(const GemmArgs& a_in, hipStream_t stream) {
  GemmArgs a = a_in;
  // Row-major inside the band for every shape of the model (NT <= 16: measured neutral, profiles/r06_v33_gemm_walk.log); problems
  // with 24 or more tile columns walk groups of 8 tile rows (8192^3: 1 309 -> 1 397 TFLOP/s - with one tile row per XCD round the
  // 33 operand panels of a K step do not stay in the 4 MB L2)
  a.walk_gm = g_gemm8p_walk >= 1 ? g_gemm8p_walk : ((a.N + 255) / 256 >= 24 ? 8 : 1);
  if (a.conv_c > 0 && a.conv_korder == 1) {  // slab-major VGG layers: the patch-resident kernel (conv_patch.hip, round 6)
    const int rc = conv_patch_try_launch(a, stream);
    if (rc <= 0) return rc;
  }
  if (a.conv_c == 64 || a.conv_c == 128) {  // the three widest VGG layers have their own kernels (conv64.hip)
    const int rc = conv64_try_launch(a, stream);
    if (rc <= 0) return rc;
  }
  static const int use_env = getenv("ROMA_GEMM8P") ? atoi(getenv("ROMA_GEMM8P")) : 1;
  const int use = g_gemm_tuning[0] >= 0 ? g_gemm_tuning[0] : use_env;
  if (!use) return 1;
  if ((long)a.M * 1 >= (1L << 31) - 512) return 1;
  if (a.in_dt != DT_BF16 || a.batch != 1 || a.batch2 != 1 || a.lower_only || a.alpha != 1.0f) return 1;
  if (a.K % 64 != 0 || a.K < 4 * 64 || a.K > 32704) return 1;
  const bool conv = a.conv_c > 0;
  if (conv && (a.conv_c % 64 != 0 || a.conv_c > 512)) return 1;
#ifdef ROMA_R8_BUFLDS
  // the dense operands are addressed with 32-bit byte offsets below 2^31 (the out-of-range marker) inside a buffer descriptor
  if (!conv && (((long)a.M * a.lda + a.K) * 2 >= (1l << 31) || ((long)a.N * a.ldw + a.K) * 2 >= (1l << 31))) return 1;
#endif
  // shapes gemm.hip would run on 256 x 256 tiles; for a single pair (M = 3 202 token rows, BASELINE config 2) the wide
  // launches (qkv, fc1: N >= 2048, 150-210 tiles of 256 x 256) are still better off on this kernel's pipelined loop with
  // part of the CUs idle than on gemm.hip's 128 x 128 loop at one wave per SIMD (ROMA_GEMM8P_MINM: A/B)
  static const long minm_env = getenv("ROMA_GEMM8P_MINM") ? atol(getenv("ROMA_GEMM8P_MINM")) : 2048;
  const bool big_m = (long)a.M >= 8192 || ((long)a.M >= minm_env && a.N >= 2048 && !conv);
  if (!big_m) return 1;
  bool tile256 = false;
  if (a.N >= 384) {
    const long w256 = ((a.N + 255) / 256) * 256, w192 = ((a.N + 191) / 192) * 192;
    tile256 = !(w192 < w256);
    if (!tile256) {
      const int rc = ws1x1_try_launch(a, stream);  // C = 576 refiner 1x1: weight-stationary (ws1x1.hip)
      if (rc <= 0) return rc;
      return gemm6p_try_launch(a, stream);  // 256 x 192 tiles (gemm6p.hip)
    }
  } else if (a.N > 192 && a.N <= 256) {
    tile256 = true;
  }
  if (!tile256) return 1;
  if ((reinterpret_cast<uintptr_t>(a.C) & 15) != 0) return 1;
  if (a.out_dt == DT_BF16) {
    if (a.mode == EPI_QKV) {
      if (!a.qkv_pad || conv || (a.heads * a.hd) % 64 != 0) return 1;
      return launch8p<bf16_t, false, E8_QKV>(a, stream, "qkv");
    }
    if (a.mode != EPI_STD || a.res != nullptr || (a.ldc & 7) != 0) return 1;
    if (a.scale) return 1;  // a per-column scale stays on gemm.hip (the model folds LayerScale into the weights instead)
    if (a.res_bf16) {
      if (conv || a.act != ACT_NONE) return 1;
      return launch8p<bf16_t, false, E8_RESBF16>(a, stream, "res_bf16");
    }
    if (conv) {
      if (a.act != ACT_RELU) return 1;
      return launch8p<bf16_t, true, E8_RELU>(a, stream, "relu");
    }
    if (a.act == ACT_GELU) return launch8p<bf16_t, false, E8_GELU>(a, stream, "gelu");
    if (a.act == ACT_RELU) return launch8p<bf16_t, false, E8_RELU>(a, stream, "relu");
#ifdef ROMA_TOOLS_BUILD
    // A/B switch: LDS-DMA issued between the MFMAs instead of in the load block.  Measured (tools/bench_gemm_overhead.py bit 512, profiles/r02_v11_gemm_overhead.log):
    // neutral at K = 1024, -4 % at K = 4096, -15 % on the 3x3 convolution (its per-piece select sits between the MFMAs)
    if (a.dbg & 512) return launch8p_s<bf16_t, false, E8_NONE, true>(a, stream, "none");
    if (a.dbg & 32768) {  // ablation / trace builds (tools/bench_gemm_ablation.py)
      const int sched = gemm8p_sched_of(a);
      const int abl = 1 | ((a.dbg & 4096) ? 2 : 0) | ((a.dbg & 8192) ? 4 : 0) | ((a.dbg & 16384) ? 8 : 0);
#define R8_ABL_CASE(S, X) \
  if (sched == S && abl == X) return launch8p_s<bf16_t, false, E8_NONE, false, S, X>(a, stream, "none");
      R8_ABL_CASE(0, 1) R8_ABL_CASE(0, 3) R8_ABL_CASE(0, 5) R8_ABL_CASE(0, 9) R8_ABL_CASE(0, 7) R8_ABL_CASE(0, 11) R8_ABL_CASE(0, 13)
      R8_ABL_CASE(1, 1) R8_ABL_CASE(1, 3) R8_ABL_CASE(1, 5) R8_ABL_CASE(1, 9) R8_ABL_CASE(1, 7) R8_ABL_CASE(1, 11) R8_ABL_CASE(1, 13)
#undef R8_ABL_CASE
      set_error("gemm8p: no such ablation build");
      return -2;
    }
#else
    ROMA_REQUIRE(!(a.dbg & (512 | 32768)), "gemm8p: the DMA-between-MFMAs variant and the ablation / trace builds exist in tools builds only (make TOOLS=1)");
#endif
    return launch8p<bf16_t, false, E8_NONE>(a, stream, "none");
  }
  if (a.out_dt == DT_F32) {
    if (conv || a.mode != EPI_STD || a.act == ACT_GELU || a.res_bf16) return 1;
    if ((a.ldc & 3) != 0 || (a.N & 3) != 0) return 1;
    if (a.res && ((a.ldr & 3) != 0 || (reinterpret_cast<uintptr_t>(a.res) & 15) != 0)) return 1;
    if (a.act == ACT_RELU) return launch8p<float, false, E8_RELU>(a, stream, "relu");
    return launch8p<float, false, E8_NONE>(a, stream, "none");
  }
  return 1;
}

}  // namespace roma
