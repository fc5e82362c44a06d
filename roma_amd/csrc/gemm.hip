// MFMA GEMM for gfx950 (see gemm.h).  Workgroups of 4 wave64s (small / HBM-bound tiles, several per CU) or 8 wave64s
// (256-row tiles: one persistent workgroup per CU that walks its XCD's band of tiles).
//
// Tile anatomy (CDNA4):
//   * K is staged in slabs of 8 x 16-byte chunks per row (32 f32 / 64 bf16 = 128-byte LDS rows) with
//     `global_load_lds_dwordx4` (direct HBM->LDS DMA, no VGPR round trip, no ds_write pass).  The DMA
//     writes lane-linear (wave-uniform base + lane*16 B), so the bank-conflict fix is an XOR swizzle applied
//     to the per-lane SOURCE address and again on the ds_read side (same involution on both):
//         LDS slot(row, chunk) = chunk ^ ((row >> 1) & 7)
//     a ds_read_b128 lane group then touches 16 distinct 16-B slots of the 256-B bank row: conflict free.
//   * out-of-range rows / K tail / 3x3-conv zero padding are DMA'd from a 256-byte zero page, so every
//     tail is exact without predicated stores into LDS.
//   * two LDS buffers, ONE raw `s_barrier` per slab: the DMA of slab t+1 runs under slab t's MFMAs (8-wave tiles
//     issue it first and read their fragments with inline asm, 4-wave tiles issue it after the fragment reads:
//     hipcc drains the DMA queue before any LDS read it can see).
//   * MFMA operand roles are SWAPPED: the weight tile feeds the A operand (rows = n) and the
//     activation tile the B operand (cols = m).  D[n][m] then puts 4 CONSECUTIVE n of one output
//     row m in each lane's register quad, so bias/scale/residual/output move as 16-byte vectors.
//   * f32 path: v_mfma_f32_32x32x2_f32 (exact f32); the k-order inside a slab is permuted (lane half h
//     takes k = 8g+4h+s) which is legal because A and B use the same permutation.
#include "gemm.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

namespace roma {

}  // namespace roma

#include "gemm_device.h"

namespace roma {

template <typename TIN, typename TOUT, int WM, int WN, int TM, int TN, bool CONV>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_kernel(const GemmArgs a) {
  constexpr int CE = InTraits<TIN>::CE;
  constexpr int BKE = 8 * CE;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int NWAVES = WM * WN;
  constexpr int NA = BM / (8 * NWAVES), NW = BN / (8 * NWAVES);  // DMA instructions per wave per slab (8 rows x 128 B each)
  static_assert(BM % (8 * NWAVES) == 0 && BN % (8 * NWAVES) == 0, "tile rows must split evenly over the waves");
  constexpr int BUF = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int bz = blockIdx.z;
  // XCD-aware tile order (the dispatcher places workgroup b on XCD b % 8, each XCD has a private L2): every XCD
  // walks a contiguous band of m-tiles with n fastest, so the n-tiles sharing one activation tile - and, for the
  // 3x3 convolution, the vertically adjacent image rows - hit the same L2 instead of re-reading HBM 8 times.
  const int NT = (a.N + BN - 1) / BN;
  const long nblk = (long)((a.M + BM - 1) / BM) * NT;
  const long per_xcd = (nblk + 7) / 8;
  // Persistent workgroups: each walks the tiles li, li + wg_per_xcd, ... of its XCD band.  While the last K slab
  // of a tile is multiplied the FIRST slab of the workgroup's next tile is already DMA'd, and it keeps flying
  // under the epilogue (which stages through its own LDS slice), so launch latency, the cold first slab and the
  // epilogue are no longer serialised per tile (they cost ~15 us per 256x256 tile with one workgroup per CU).
  const int xcd = blockIdx.x % 8;
  const long wg_per_xcd = gridDim.x / 8;
  long li = blockIdx.x / 8;
  if (li >= per_xcd || (long)xcd * per_xcd + li >= nblk) return;

  const int bz1 = bz / a.batch2, bz2 = bz - bz1 * a.batch2;  // two batch levels (batch2 = 1: bz1 = bz, bz2 = 0)
  const TIN* Ab = reinterpret_cast<const TIN*>(a.A) + (long)bz1 * a.sA + (long)bz2 * a.sA2;
  const TIN* Wb = reinterpret_cast<const TIN*>(a.W) + (long)bz1 * a.sW + (long)bz2 * a.sW2;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // ---- per-lane DMA descriptors: lane -> (row = 8q + lane/8, slot = lane%8), source chunk = slot ^ swizzle(row)
  const char* a_src[NA];
  int a_chunk[NA], a_y[NA], a_x[NA];
  const char* w_src[NW];
  int w_chunk[NW];
  long d_m0 = 0;  // tile the descriptors currently describe
  int d_n0 = 0;
#define ROMA_TILE_SETUP(LTILE)                                                                              \
  {                                                                                                         \
    d_m0 = ((LTILE) / NT) * BM;                                                                             \
    d_n0 = (int)((LTILE) % NT) * BN;                                                                        \
    int ln_ = lane;                                                                                         \
    asm volatile("" : "+v"(ln_)); /* opaque: each setup is computed where it stands (see tile top) */         \
    const int r8 = ln_ >> 3, slot = ln_ & 7;                                                                \
    _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                                        \
      const int row = 8 * (wave * NA + j) + r8;                                                             \
      const int chunk = slot ^ ((row >> 1) & 7);                                                            \
      const long gm = d_m0 + row;                                                                           \
      a_chunk[j] = chunk;                                                                                   \
      if (gm < a.M) {                                                                                       \
        if (CONV) {                                                                                         \
          const long hw = (long)a.conv_h * a.conv_w;                                                        \
          const long b = gm / hw;                                                                           \
          const int rem = (int)(gm - b * hw);                                                               \
          a_y[j] = rem / a.conv_w;                                                                          \
          a_x[j] = rem - a_y[j] * a.conv_w;                                                                 \
          a_src[j] = reinterpret_cast<const char*>(Ab + gm * a.conv_c + chunk * CE);                        \
        } else if (a.qkv_pad) { /* rows = (image, padded token); tokens >= ntok read the zero page */      \
          const long qb_ = gm / a.npad;                                                                     \
          const int qt_ = (int)(gm - qb_ * a.npad);                                                         \
          a_y[j] = a_x[j] = 0;                                                                              \
          a_src[j] = qt_ < a.ntok ? reinterpret_cast<const char*>(Ab + (qb_ * a.ntok + qt_) * a.lda + chunk * CE) : nullptr; \
        } else {                                                                                            \
          a_y[j] = a_x[j] = 0;                                                                              \
          a_src[j] = reinterpret_cast<const char*>(Ab + gm * a.lda + chunk * CE);                           \
        }                                                                                                   \
      } else {                                                                                              \
        a_y[j] = -100000; /* conv: every tap out of range */                                                \
        a_x[j] = 0;                                                                                         \
        a_src[j] = nullptr;                                                                                 \
      }                                                                                                     \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                                        \
      const int row = 8 * (wave * NW + j) + r8;                                                             \
      const int chunk = slot ^ ((row >> 1) & 7);                                                            \
      w_chunk[j] = chunk;                                                                                   \
      w_src[j] = (d_n0 + row < a.N) ? reinterpret_cast<const char*>(Wb + (long)(d_n0 + row) * a.ldw + chunk * CE) : nullptr; \
    }                                                                                                       \
  }

#define ROMA_ISSUE_SLAB_A(KT, BUFI)                                                                         \
  {                                                                                                         \
    const int k0_ = (KT) * BKE;                                                                             \
    char* abuf_ = smem + (BUFI) * BUF;                                                                      \
    if (CONV) {                                                                                             \
      const int tap_ = k0_ / a.conv_c;                                                                      \
      const int c0_ = k0_ - tap_ * a.conv_c;                                                                \
      const int dy_ = tap_ / 3 - 1, dx_ = tap_ % 3 - 1;                                                     \
      const long toff_ = (((long)dy_ * a.conv_w + dx_) * a.conv_c + c0_) * (long)sizeof(TIN);               \
      _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                                      \
        const int yy_ = a_y[j] + dy_, xx_ = a_x[j] + dx_;                                                   \
        const bool ok_ = yy_ >= 0 && yy_ < a.conv_h && xx_ >= 0 && xx_ < a.conv_w;                         \
        glds16(ok_ ? a_src[j] + toff_ : zero, abuf_ + (wave * NA + j) * 1024);                              \
      }                                                                                                     \
    } else {                                                                                                \
      _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                                      \
        const bool ok_ = a_src[j] != nullptr && (k0_ + a_chunk[j] * CE < a.K);                              \
        glds16(ok_ ? a_src[j] + (long)k0_ * sizeof(TIN) : zero, abuf_ + (wave * NA + j) * 1024);            \
      }                                                                                                     \
    }                                                                                                       \
  }
#define ROMA_ISSUE_SLAB_W(KT, BUFI)                                                                         \
  {                                                                                                         \
    const int k0_ = (KT) * BKE;                                                                             \
    char* wbuf_ = smem + (BUFI) * BUF + BM * ROWB;                                                          \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                                        \
      const bool ok_ = w_src[j] != nullptr && (k0_ + w_chunk[j] * CE < a.K);                                \
      glds16(ok_ ? w_src[j] + (long)k0_ * sizeof(TIN) : zero, wbuf_ + (wave * NW + j) * 1024);              \
    }                                                                                                       \
  }
#define ROMA_ISSUE_SLAB(KT, BUFI) \
  ROMA_ISSUE_SLAB_A(KT, BUFI)     \
  ROMA_ISSUE_SLAB_W(KT, BUFI)

  // fragment read offsets: row = tile_row0 + l31 (tile_row0 % 32 == 0), slot = (2g + h) ^ ((l31 >> 1) & 7)
  const int sw = (l31 >> 1) & 7;
  int rd_off[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) rd_off[g] = l31 * ROWB + (((2 * g + h) ^ sw) << 4);

  const int nk = (a.dbg & 2) ? 1 : (a.K + BKE - 1) / BKE;
  if constexpr (NWAVES == 8) {
    // Persistent equal-sized tiles keep every CU in lock-step: all epilogues hit HBM in one write burst while no
    // MFMA runs, then HBM idles.  Start every other workgroup half a tile late so the two phases interleave chip-wide.
    if ((a.dbg & 16) && ((blockIdx.x >> 3) & 1) && nblk > gridDim.x) {
      const long t0 = __builtin_readcyclecounter();
      const long wait = (long)nk * 1100;
      while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
  }
  ROMA_TILE_SETUP((long)xcd * per_xcd + li);
  ROMA_ISSUE_SLAB(0, 0);
  int bsel = 0;  // LDS buffer holding slab 0 of the current tile
  for (bool first_tile = true;; first_tile = false) {
  if constexpr (NWAVES == 8) {
    // The descriptors of this tile were built (and used for its slab 0) under the last MFMAs of the previous tile; they
    // are rebuilt here instead of being kept: 4 x (NA + NW) registers live across the epilogue pushed its preloaded
    // bias / scale / residual columns into scratch (the 256 x 192 tile went from 247 to 335 us on the stride-8 refiner
    // GEMM).  ~40 integer instructions per tile.
    if (!first_tile) ROMA_TILE_SETUP((long)xcd * per_xcd + li);
  }
  const long m0 = d_m0;
  const int n0 = d_n0;
  const long li_next = li + wg_per_xcd;
  const bool has_next = li_next < per_xcd && (long)xcd * per_xcd + li_next < nblk;
  const bool skip_tile = a.lower_only && n0 > m0 + BM - 1;  // (lower_only launches are never persistent)

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One barrier per slab: [slab kt landed] -> barrier -> DMA of slab kt+1 (or of the next tile's slab 0) into the
  // other buffer, which every wave finished reading before it passed this barrier -> fragments + MFMAs of slab kt
  // under that DMA.  The fragment reads are inline asm: hipcc drains the DMA queue (vmcnt(0)) before any ds_read
  // it can see (the read may alias an LDS-DMA target), which would serialise DMA and math; hidden reads also let
  // the fragments be fetched per k-group (2 x (TM+TN) registers, double buffered with counted lgkmcnt) instead of
  // holding the whole slab - the registers that the cross-tile prefetch state needs.
#define ROMA_READ_G(WV, AV, G)                                                                              \
  {                                                                                                         \
    const unsigned wa_ = sb + woff + rd_off[G], aa_ = sb + aoff + rd_off[G];                                \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(WV[tn]) : "v"(wa_), "n"(tn * 32 * ROWB));       \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(AV[tm]) : "v"(aa_), "n"(tm * 32 * ROWB));       \
  }
#define ROMA_MFMA_G(WV, AV)                                                                                 \
  _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) {     \
    if constexpr (sizeof(TIN) == 4) {                                                                       \
      acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(WV[tn].x), __uint_as_float(AV[tm].x), acc[tn][tm], 0, 0, 0); \
      acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(WV[tn].y), __uint_as_float(AV[tm].y), acc[tn][tm], 0, 0, 0); \
      acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(WV[tn].z), __uint_as_float(AV[tm].z), acc[tn][tm], 0, 0, 0); \
      acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(WV[tn].w), __uint_as_float(AV[tm].w), acc[tn][tm], 0, 0, 0); \
    } else {                                                                                                \
      acc[tn][tm] = mfma_h16_32x32x16(WV[tn],           \
                                                            AV[tm], acc[tn][tm]); \
    }                                                                                                       \
  }
#define ROMA_WAIT_LGKM(N)                                  \
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); \
  __builtin_amdgcn_sched_barrier(0);
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned aoff = (wm * TM) * 32 * ROWB, woff = BM * ROWB + (wn * TN) * 32 * ROWB;
  // 8-wave tiles: the LAST k-group of a slab is multiplied only after the next slab's barrier (its fragments are
  // carried in wvB / avB, zero for the first slab of a tile): those 8 MFMAs per wave are queued the moment the
  // barrier opens and cover the DMA issue and the first fragment reads of the new slab, during which both waves of
  // a SIMD used to leave the MFMA pipe idle (SQ counters: 39 % of the wave cycles parked).
  // (bf16 only: the exact-f32 parity kernels keep the in-place order - with the carry they produced wrong sums, most
  //  likely asynchronous asm-read fragments passing through a compiler-made copy; not worth chasing for that mode)
  constexpr bool CARRY = sizeof(TIN) == 2;
  uint4 wvA[TN], avA[TM], wvB[TN], avB[TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) wvB[tn] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) avB[tm] = make_uint4(0, 0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = (bsel + kt) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (NWAVES == 8) {
      if constexpr (CARRY) {
        ROMA_MFMA_G(wvB, avB);
        __builtin_amdgcn_sched_barrier(0);
      }
      // large MFMA-bound tiles: DMA first, fragments per k-group from inline asm (see above)
      // the next slab's DMA is issued in two bursts (A rows now, W rows after the first MFMA group): both waves of
      // a SIMD leave the barrier together, and a single 8-piece burst per wave kept the MFMA pipe idle behind the
      // VMEM issue (~60-180 cycles per piece)
      int ikt = -1;
      if (kt + 1 < nk) {
        if (!(a.dbg & 4)) ikt = kt + 1;
      } else if (has_next) {
        ROMA_TILE_SETUP((long)xcd * per_xcd + li_next);
        ikt = 0;
      }
      const bool split = !(a.dbg & 32);
      if (ikt >= 0) {
        ROMA_ISSUE_SLAB_A(ikt, cur ^ 1);
        if (!split) ROMA_ISSUE_SLAB_W(ikt, cur ^ 1);
      }
      if (skip_tile || (a.dbg & 8)) {
        if (ikt >= 0 && split) ROMA_ISSUE_SLAB_W(ikt, cur ^ 1);
        continue;
      }
      const unsigned sb = lds0 + cur * BUF;
      ROMA_READ_G(wvA, avA, 0);
      ROMA_READ_G(wvB, avB, 1);
      ROMA_WAIT_LGKM(TN + TM);
      ROMA_MFMA_G(wvA, avA);
      __builtin_amdgcn_sched_barrier(0);
      if (ikt >= 0 && split) ROMA_ISSUE_SLAB_W(ikt, cur ^ 1);
      ROMA_READ_G(wvA, avA, 2);
      ROMA_WAIT_LGKM(TN + TM);
      ROMA_MFMA_G(wvB, avB);
      __builtin_amdgcn_sched_barrier(0);
      ROMA_READ_G(wvB, avB, 3);
      ROMA_WAIT_LGKM(TN + TM);
      ROMA_MFMA_G(wvA, avA);
      ROMA_WAIT_LGKM(0);
      if constexpr (!CARRY) { ROMA_MFMA_G(wvB, avB); }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // small / HBM-bound tiles (4 waves, several workgroups per CU): compiler-scheduled reads of the whole slab,
      // then the DMA of the next slab (hipcc would drain the DMA queue before a visible ds_read otherwise), then MFMAs
      const char* As = smem + cur * BUF;
      const char* Ws = As + BM * ROWB;
      uint4 wv[4][TN], av[4][TM];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          wv[g][tn] = *reinterpret_cast<const uint4*>(Ws + (wn * TN + tn) * 32 * ROWB + rd_off[g]);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          av[g][tm] = *reinterpret_cast<const uint4*>(As + (wm * TM + tm) * 32 * ROWB + rd_off[g]);
      }
      if (kt + 1 < nk) {
        ROMA_ISSUE_SLAB(kt + 1, cur ^ 1);
      } else if (has_next) {
        ROMA_TILE_SETUP((long)xcd * per_xcd + li_next);
        ROMA_ISSUE_SLAB(0, cur ^ 1);
      }
      if (skip_tile) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) ROMA_MFMA_G(wv[g], av[g]);
    }
  }
  if constexpr (NWAVES == 8 && CARRY) {
    ROMA_MFMA_G(wvB, avB);  // the carried last k-group of the tile
  }
#undef ROMA_WAIT_LGKM
#undef ROMA_MFMA_G
#undef ROMA_READ_G
  bsel = (bsel + nk) & 1;
  li = li_next;
  if (skip_tile) {
    if (!has_next) break;
    continue;
  }

  // ---------------------------------------------------------------- epilogue
  // opaque lane id: every lane-dependent address of the epilogue variants is computed HERE, per tile.  With the plain
  // `lane` hipcc hoists them all (every inlined variant's) to kernel entry and parks ~60 registers in scratch across the
  // K loop - and reloads some inside it (tools/kernel_resources.py: 169 spilled registers on the 256 x 192 bf16 tile).
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l31e = lane_e & 31, he = lane_e >> 5;
  TOUT* Cb = reinterpret_cast<TOUT*>(a.C) + (long)bz1 * a.sC + (long)bz2 * a.sC2;
  const float* Rb = a.res ? a.res + (long)bz1 * a.sR + (long)bz2 * a.sR2 : nullptr;
  const bool vecC = ((a.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0);
  const bool vecR = Rb && ((a.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(Rb) & 15) == 0);

  {
    // staged plain epilogues (see epi_staged_*).  8-wave (persistent) tiles stage in their own slices behind the
    // operand buffers (the next tile's first slab is already being DMA'd into those); 4-wave tiles reuse the
    // operand buffers after a barrier (keeps 2+ workgroups per CU).
    constexpr int SLICE = sizeof(TOUT) == 2 ? 32 * TN * 64 : 4096;
    if constexpr (sizeof(TOUT) == 2) {
      if (a.mode == EPI_QKV && a.qkv_pad && ((a.heads * a.hd) % (TN * 32)) == 0) {
        if constexpr (NWAVES != 8) __builtin_amdgcn_s_barrier();
        epi_staged_qkv<TM, TN>(acc, a, smem + (NWAVES == 8 ? 2 * BUF : 0) + wave * SLICE, m0 + (long)wm * TM * 32,
                               n0 + wn * TN * 32, lane_e);
        if (!has_next) break;
        continue;  // next tile
      }
    }
    if constexpr (sizeof(TOUT) == 4 && TM * TN <= 6) {  // (not the 256 x 256 tile: it is at 256 registers and COSK problems never take it)
      if (a.mode == EPI_COSK && vecC && (a.N & 3) == 0 && (a.sNy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.ny) & 15) == 0) {
        if constexpr (NWAVES != 8) __builtin_amdgcn_s_barrier();
        epi_staged_cosk<TM, TN>(acc, a, reinterpret_cast<float*>(Cb), a.nx + (long)bz * a.sNx, a.ny + (long)bz * a.sNy,
                                smem + (NWAVES == 8 ? 2 * BUF : 0) + wave * SLICE, m0 + (long)wm * TM * 32, n0 + wn * TN * 32, lane_e);
        if (!has_next) break;
        continue;  // next tile
      }
    }
    bool staged = a.mode == EPI_STD && (reinterpret_cast<uintptr_t>(Cb) & 15) == 0;
    if constexpr (sizeof(TOUT) == 2)  // res_bf16: see gemm_launch; no per-column scale in the bf16 row writer
      staged = staged && Rb == nullptr && (a.ldc & 7) == 0 && (a.scale == nullptr || a.res_bf16 != nullptr);
    else staged = staged && vecC && (Rb == nullptr || vecR) && (a.N & 3) == 0 && a.act != ACT_GELU;
    if (staged) {
      if constexpr (NWAVES != 8) __builtin_amdgcn_s_barrier();
      char* ws = smem + (NWAVES == 8 ? 2 * BUF : 0) + wave * SLICE;
      const long mw0 = m0 + (long)wm * TM * 32;
      const int nw0 = n0 + wn * TN * 32;
      const bool full_tile = m0 + BM <= a.M && n0 + BN <= a.N;
      if constexpr (sizeof(TOUT) == 2) {
        bf16_t* Cbb = reinterpret_cast<bf16_t*>(Cb);
        if (a.act == ACT_GELU) {
          if (full_tile) epi_staged_bf16<TM, TN, ACT_GELU, true>(acc, a, Cbb, ws, mw0, nw0, lane_e);
          else epi_staged_bf16<TM, TN, ACT_GELU, false>(acc, a, Cbb, ws, mw0, nw0, lane_e);
        } else if (a.act == ACT_RELU) {
          if (full_tile) epi_staged_bf16<TM, TN, ACT_RELU, true>(acc, a, Cbb, ws, mw0, nw0, lane_e);
          else epi_staged_bf16<TM, TN, ACT_RELU, false>(acc, a, Cbb, ws, mw0, nw0, lane_e);
        } else if (a.res_bf16) {
          const bf16_t* Rbb = reinterpret_cast<const bf16_t*>(a.res_bf16) + (long)bz * a.sR;
          if (a.scale) {  // operator entry point only (the model folds LayerScale into the weights)
            epi_staged_bf16<TM, TN, ACT_NONE, false, true, true>(acc, a, Cbb, ws, mw0, nw0, lane_e, Rbb);
          } else {
            if (full_tile) epi_staged_bf16<TM, TN, ACT_NONE, true, true>(acc, a, Cbb, ws, mw0, nw0, lane_e, Rbb);
            else epi_staged_bf16<TM, TN, ACT_NONE, false, true>(acc, a, Cbb, ws, mw0, nw0, lane_e, Rbb);
          }
        } else {
          if (full_tile) epi_staged_bf16<TM, TN, ACT_NONE, true>(acc, a, Cbb, ws, mw0, nw0, lane_e);
          else epi_staged_bf16<TM, TN, ACT_NONE, false>(acc, a, Cbb, ws, mw0, nw0, lane_e);
        }
      } else {
        float* Cbf = reinterpret_cast<float*>(Cb);
        if (a.act == ACT_RELU) {
          if (full_tile) epi_staged_f32<TM, TN, ACT_RELU, true>(acc, a, Cbf, Rb, ws, mw0, nw0, lane_e);
          else epi_staged_f32<TM, TN, ACT_RELU, false>(acc, a, Cbf, Rb, ws, mw0, nw0, lane_e);
        } else {
          if (full_tile) epi_staged_f32<TM, TN, ACT_NONE, true>(acc, a, Cbf, Rb, ws, mw0, nw0, lane_e);
          else epi_staged_f32<TM, TN, ACT_NONE, false>(acc, a, Cbf, Rb, ws, mw0, nw0, lane_e);
        }
      }
      if (!has_next) break;
      continue;  // next tile
    }
  }

#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const long m = m0 + (wm * TM + tm) * 32 + l31e;
    if (m >= a.M) continue;
    float nxm = 0.f;
    if (a.mode == EPI_COSK) nxm = a.nx[(long)bz * a.sNx + m];
    int qb = 0, qt = 0;
    if (a.mode == EPI_QKV) {
      const int per = a.qkv_pad ? a.npad : a.ntok;
      qb = (int)(m / per);
      qt = (int)(m - (long)qb * per);
      if (qt >= a.ntok) continue;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + (wn * TN + tn) * 32 + 8 * rg + 4 * he;
        const int nvalid = a.N - n;
        if (nvalid <= 0) continue;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a.alpha * acc[tn][tm][4 * rg + j];
        if (a.mode == EPI_COSK) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < nvalid) {
              const float nyn = a.ny[(long)bz * a.sNy + n + j];
              float c = v[j] / (nxm * nyn + 1e-6f);
              float kk = expf((c - 1.0f) * a.inv_t);
              if (a.diag_add != 0.f && m == n + j) kk += a.diag_add;
              v[j] = kk;
            }
          }
          store4<TOUT>(Cb + m * a.ldc + n, v, vecC, nvalid);
          continue;
        }
        if (a.bias) {
          if (nvalid >= 4) {
            v += *reinterpret_cast<const f32x4*>(a.bias + n);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < nvalid) v[j] += a.bias[n + j];
          }
        }
        if (a.mode == EPI_QKV) {
          const int D = a.heads * a.hd;
          const int which = n / D;
          const int rem = n - which * D;
          const int head = rem / a.hd;
          const int d = rem - head * a.hd;
          const long bh = (long)qb * a.heads + head;
          if (which == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= a.qscale;
            ElemIO<TOUT>::st4(reinterpret_cast<TOUT*>(a.q) + (bh * a.npad + qt) * a.hd + d, v);
          } else if (which == 1) {
            ElemIO<TOUT>::st4(reinterpret_cast<TOUT*>(a.k) + (bh * a.npad + qt) * a.hd + d, v);
          } else {
            TOUT* vp = reinterpret_cast<TOUT*>(a.vt) + (bh * a.hd + d) * a.npad + qt;
#pragma unroll
            for (int j = 0; j < 4; ++j) ElemIO<TOUT>::st(vp + (long)j * a.npad, v[j]);
          }
          continue;
        }
        if (a.act == ACT_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        } else if (a.act == ACT_GELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
        }
        if (a.scale) {
          if (nvalid >= 4) {
            v *= *reinterpret_cast<const f32x4*>(a.scale + n);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < nvalid) v[j] *= a.scale[n + j];
          }
        }
        if (Rb) {
          if (vecR && nvalid >= 4) {
            v += *reinterpret_cast<const f32x4*>(Rb + m * a.ldr + n);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < nvalid) v[j] += Rb[m * a.ldr + n + j];
          }
        }
        store4<TOUT>(Cb + m * a.ldc + n, v, vecC, nvalid);
      }
    }
  }
  if (!has_next) break;
  }  // tile loop
#undef ROMA_ISSUE_SLAB
#undef ROMA_ISSUE_SLAB_A
#undef ROMA_ISSUE_SLAB_W
#undef ROMA_TILE_SETUP
}

template <typename TIN, typename TOUT, int WM, int WN, int TM, int TN, bool CONV>
static int launch_cfg(const GemmArgs& a, hipStream_t stream) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const long nblk = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  size_t lds = (size_t)2 * (BM + BN) * ROWB;
  {  // epilogue staging slices (one per wave): bf16 rows of the wave tile, or one 32 x 32 f32 block
    const size_t stg = (size_t)WM * WN * (sizeof(TOUT) == 2 ? 32 * TN * 64 : 4096);
    lds = (WM * WN == 8) ? lds + stg : std::max(lds, stg);
  }
  // persistent grid: the co-resident workgroups (256 CUs x LDS-limited occupancy), a multiple of the 8 XCDs;
  // lower_only (skipped tiles) and batched launches keep one tile per workgroup
  long gx = ((nblk + 7) / 8) * 8;
  // (the 4-wave tiles serve the small / HBM-bound problems: several short-lived workgroups per CU overlap their
  //  store drain with each other better than one persistent workgroup that waits on its own stores)
  if (!a.lower_only && a.batch * a.batch2 == 1 && WM * WN == 8) {
    const long occ = std::max<long>(1, std::min<long>(8, (160 * 1024) / (long)lds));
    gx = std::min<long>(gx, 256 * occ);
  }
  dim3 grid((unsigned)gx, 1, (unsigned)(a.batch * a.batch2));
  char pname[96];
  snprintf(pname, sizeof pname, "gemm_kernel<%s,%s,%d,%d,%d,%d,%s>", sizeof(TIN) == 4 ? "f32" : ROMA_H16_NAME,
           sizeof(TOUT) == 4 ? "f32" : ROMA_H16_NAME, WM, WN, TM, TN, CONV ? "conv3x3" : "dense");
  // algorithmic FLOPs: the caller's M (a.M may have been padded to npad tokens per image for the QKV epilogue)
  ProfScope ps(pname, 2.0 * (double)(a.m_alg > 0 ? a.m_alg : a.M) * (a.n_alg > 0 ? a.n_alg : a.N) * (a.k_alg > 0 ? a.k_alg : a.K) * a.batch * a.batch2 *
                          (a.lower_only ? 0.5 : 1.0), "flop", stream);
  // the > 64 KB dynamic-LDS opt-in is a per-DEVICE function attribute: one flag per device ordinal
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<TIN, TOUT, WM, WN, TM, TN, CONV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_kernel<TIN, TOUT, WM, WN, TM, TN, CONV>), grid, dim3(WM * WN * 64), lds, stream, a);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// Tile selection.  The DMA'd operand traffic is 2*(BM+BN)*K bytes per BM*BN*K MACs, and ~13 TB/s of L2->LDS
// traffic is what the chip sustains, so the large MFMA-bound problems need the 256-row tiles (8 waves, 128 KB LDS):
// 128x128 caps at ~0.4-0.6 PF/s (64 FLOP/B), 256x192 / 256x256 at 110 / 128 FLOP/B.
template <typename TIN, typename TOUT, bool CONV>
static int launch_shape(const GemmArgs& a, hipStream_t stream) {
  if (a.N <= 32) return launch_cfg<TIN, TOUT, 4, 1, 1, 1, CONV>(a, stream);       // 128 x 32
  const bool big_m = (long)a.M * a.batch * a.batch2 >= 8192 && a.M >= 1024;
  if (big_m && a.N >= 384) {
    const long w256 = ((a.N + 255) / 256) * 256, w192 = ((a.N + 191) / 192) * 192;
    if (w192 < w256) return launch_cfg<TIN, TOUT, 4, 2, 2, 3, CONV>(a, stream);    // 256 x 192
    return launch_cfg<TIN, TOUT, 2, 4, 4, 2, CONV>(a, stream);                     // 256 x 256
  }
  if (big_m && a.N > 192 && a.N <= 256) return launch_cfg<TIN, TOUT, 2, 4, 4, 2, CONV>(a, stream);  // 256 x 256
  if (a.N <= 64) {
    if (big_m) return launch_cfg<TIN, TOUT, 8, 1, 1, 2, CONV>(a, stream);         // 256 x 64
    return launch_cfg<TIN, TOUT, 4, 1, 1, 2, CONV>(a, stream);                    // 128 x 64
  }
  if (big_m && a.N <= 128) return launch_cfg<TIN, TOUT, 4, 2, 2, 2, CONV>(a, stream);  // 256 x 128
  if constexpr (!CONV) {
    if (a.N > 128 && a.N <= 160) return launch_cfg<TIN, TOUT, 4, 1, 1, 5, false>(a, stream);  // 128 x 160: one n-tile
  }
  // Small-M problems (a single pair: M = 3 202 token rows): N = 1024 gives 26 x 8 = 208 tiles of 128 x 128 for 256 CUs -
  // 48 CUs idle and ONE wave per SIMD on the rest, so every DMA / LDS / MFMA latency is exposed (370 TFLOP/s,
  // profiles/r02_final_bench_coarse.json).  128 x 64 tiles double the workgroups; three fit a CU (48 KiB of LDS each), so
  // all of them are resident at once and a CU interleaves the waves of 1-2 tiles.  ROMA_GEMM_SMALLM=0 switches it off (A/B).
  static const bool smallm_env = !(getenv("ROMA_GEMM_SMALLM") && atoi(getenv("ROMA_GEMM_SMALLM")) == 0);
  const long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch * a.batch2;
  if (smallm_env && !a.lower_only && tiles128 < 320) return launch_cfg<TIN, TOUT, 4, 1, 1, 2, CONV>(a, stream);  // 128 x 64
  return launch_cfg<TIN, TOUT, 2, 2, 2, 2, CONV>(a, stream);                      // 128 x 128
}

int gemm_launch(const GemmArgs& a0, hipStream_t stream) {
  GemmArgs a = a0;
  static const int dbg_env = getenv("ROMA_GEMM_DBG") ? atoi(getenv("ROMA_GEMM_DBG")) : 0;
  a.dbg = g_gemm_tuning[1] >= 0 ? g_gemm_tuning[1] : dbg_env;
  // Non-temporal output stores in the bf16 row writer (full tiles): the store acknowledgements are what the in-order
  // vmcnt queue of the next tile's first counted waits sits behind, and streaming stores come back sooner: -4..-5 % on the
  // K = 1024 / 1152 launches, neutral at K = 576 / 4096 (profiles/r02_v11_gemm_overhead.log, bit 1024).  ROMA_GEMM_NT=0 or
  // gemm_dbg bit 2048 switch them off (A/B).
  static const bool nt_env = !(getenv("ROMA_GEMM_NT") && atoi(getenv("ROMA_GEMM_NT")) == 0);
  if (nt_env && !(a.dbg & 2048)) a.dbg |= 1024;
  const int ce = a.in_dt == DT_F32 ? 4 : 8;
  ROMA_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0 && a.batch2 > 0, "gemm: empty problem");
  ROMA_REQUIRE(a.batch2 == 1 || (a.mode == EPI_STD && !a.res_bf16 && a.conv_c == 0), "gemm: the second batch level serves plain f32 / 16-bit problems only");
  ROMA_REQUIRE(a.K % ce == 0, "gemm: K must be a multiple of the 16-byte chunk");
  ROMA_REQUIRE(a.ldw % ce == 0 && (reinterpret_cast<uintptr_t>(a.W) & 15) == 0, "gemm: W not 16-byte aligned");
  ROMA_REQUIRE((reinterpret_cast<uintptr_t>(a.A) & 15) == 0, "gemm: A not 16-byte aligned");
  ROMA_REQUIRE(a.sA % ce == 0 && a.sW % ce == 0 && a.sA2 % ce == 0 && a.sW2 % ce == 0, "gemm: batch strides must keep 16-byte alignment");
  if (a.bias) ROMA_REQUIRE((reinterpret_cast<uintptr_t>(a.bias) & 15) == 0, "gemm: bias not 16-byte aligned");
  if (a.scale) ROMA_REQUIRE((reinterpret_cast<uintptr_t>(a.scale) & 15) == 0, "gemm: scale not 16-byte aligned");
  if (a.res_bf16) {  // only the staged bf16 row-writer knows this residual: refuse anything that would bypass it
    ROMA_REQUIRE(a.out_dt == DT_BF16 && a.mode == EPI_STD && a.act == ACT_NONE && a.res == nullptr,
                 "gemm: res_bf16 needs bf16 output, EPI_STD, no activation and no f32 residual");
    ROMA_REQUIRE(a.N % 8 == 0 && a.ldc % 8 == 0 && a.ldr % 8 == 0 && a.sC % 8 == 0 && a.sR % 8 == 0 &&
                 (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.res_bf16) & 15) == 0,
                 "gemm: res_bf16 needs 16-byte aligned rows (N, ldc, ldr multiples of 8)");
  }
  const bool conv = a.conv_c > 0;
  if (conv) {
    ROMA_REQUIRE(a.conv_c % (8 * ce) == 0, "gemm(conv3x3): Cin must be a multiple of the K slab");
    ROMA_REQUIRE(a.K == 9 * a.conv_c, "gemm(conv3x3): K != 9*Cin");
  } else {
    ROMA_REQUIRE(a.lda % ce == 0, "gemm: lda must keep 16-byte alignment");
  }
  if (a.mode == EPI_QKV) {
    ROMA_REQUIRE(a.hd % 4 == 0 && a.N == 3 * a.heads * a.hd, "gemm(qkv): bad head geometry");
    ROMA_REQUIRE(a.ntok > 0 && a.M % a.ntok == 0 && a.npad >= a.ntok, "gemm(qkv): rows must be images x tokens");
    // bf16: run the GEMM over padded rows (npad tokens per image) so the epilogue can write V^T as 16-byte pieces
    if (a.out_dt == DT_BF16 && a.npad % 32 == 0 && a.hd % 8 == 0 && a.batch == 1 &&
        (reinterpret_cast<uintptr_t>(a.q) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.k) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(a.vt) & 15) == 0) {
      a.qkv_pad = 1;
      a.m_alg = a.M;
      a.M = (a.M / a.ntok) * a.npad;
    }
  }
  {
    const int r8 = gemm8p_try_launch(a, stream);
    if (r8 <= 0) return r8;
  }
#define ROMA_GEMM_DISPATCH(TIN, TOUT)                                              \
  return conv ? launch_shape<TIN, TOUT, true>(a, stream) : launch_shape<TIN, TOUT, false>(a, stream)
  if (a.in_dt == DT_F32 && a.out_dt == DT_F32) { ROMA_GEMM_DISPATCH(float, float); }
  if (a.in_dt == DT_BF16 && a.out_dt == DT_BF16) { ROMA_GEMM_DISPATCH(bf16_t, bf16_t); }
  if (a.in_dt == DT_BF16 && a.out_dt == DT_F32) { ROMA_GEMM_DISPATCH(bf16_t, float); }
#undef ROMA_GEMM_DISPATCH
  set_error("gemm: unsupported dtype combination");
  return -1;
}

}  // namespace roma
