// 256 x 192 sibling of the 8-phase bf16 GEMM (gemm8p.hip) for the ConvRefiner 1x1 convolutions of the match() path:
// N = K = 1152 (stride 8) and 576 (stride 4), M = pairs x pixels (78 400 ... 746 496 rows at batch 8).  These are 36 of
// the ~700 GEMM launches of a step but its largest single share (19 % of the kernel time on gemm.hip's 256 x 192 loop,
// profiles/r02_kernel_stats.csv); 192 divides both widths, 256 wastes 10 % / 25 % of every tile row.
//
// Same anatomy as gemm8p (persistent XCD-banded tile walk, two wave groups one barrier apart, LDS-DMA streamed across
// output tiles with counted vmcnt, inline-asm fragment reads with register-naming waits, staged row-writer epilogue in
// its own LDS slices), re-cut for a 96-column wave tile:
//
//   * 8 waves = 2 groups (wr: rows [128 wr, +128)) x 2 row halves (wm: +64 wm) x 2 column halves (wc: cols [96 wc, +96)).
//     Wave tile 64 x 96 = TM 2 x TN 3 MFMA blocks (32x32x16 bf16), 96 accumulator registers.
//   * one K tile (64) = THREE phases of 8 MFMAs, one per 32-column block nt: P1 reads A (both row blocks, all four
//     k-groups: they stay in registers for the whole K tile) + W block 0, P2 reads W block 1, P3 reads W block 2.
//     LDS image: A rows in tile order; W region nt = LDS rows [64 nt, +64) = {cols 96 wc + 32 nt + [0, 32)} of both
//     column halves, so a region is dead (re-stageable) as soon as "its" phase has passed.
//   * LDS: 2 x 56 KiB operand buffers + 8 x 6 KiB epilogue slices = 160 KiB exactly.
//   * DMA stream, 7 pieces (8 rows x 128 B each) per wave and K tile:
//         P1(s): A pieces 2,3 of s+1      P2(s): W1, W2 of s+1      P3(s): A pieces 0,1 and W0 of s+2
//     every region is re-staged >= 2 phases after its last read (group 1 runs one barrier behind group 0: its reads of
//     phase X complete before ITS MFMAs of X, which start at group 0's second barrier of X - so group 0 may overwrite
//     from the load block of X + 2).  Counted waits, each in the load block BEFORE the phase whose reads need the data
//     (wait -> barrier -> next load block reads):
//         P3(s): vmcnt(5) - all of A(s+1), W0(s+1) landed (leaves P2(s) + P3(s) in flight)
//         P1(s): vmcnt(6) - W1(s) landed        P2(s): vmcnt(7) - W2(s) landed
//     i.e. a full K tile of DMA is always in flight.  vmcnt also counts the epilogue's stores; they are older than the
//     pieces a wait leaves in flight, so the counts stay conservative.
//   * dense operands only (no conv taps, no QKV padding): 1 + 1 instructions per DMA piece.
#include "gemm.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gemm_device.h"

namespace roma {

// zero source for out-of-range rows: the per-k-tile byte offset (k * 128) is added to EVERY lane's pointer, so the zero
// "row" must be as long as the longest K row (K <= 32704 bf16).  (one copy per translation unit: no device linking)
static __device__ __attribute__((aligned(256))) unsigned int g_zero_rows6[16384];

#define R6_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define R6_DS_READ(REG, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(REG) : "v"(ADDR), "n"(OFF))

template <typename TOUT, int ACT>
__global__ __launch_bounds__(512, 2) void gemm6p_kernel(const GemmArgs a) {
  constexpr int BM = 256, BN = 192, BK = 64;
  constexpr int TILE_A = BM * ROWB, BUF = TILE_A + BN * ROWB;  // 56 KiB per K tile
  constexpr int TM = 2, TN = 3;
  constexpr int SLICE = 32 * TN * 64;  // epilogue staging per wave: 32 rows of the wave's 96 bf16 columns
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wm = (wave >> 1) & 1, wc = wave & 1;
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
  const char* zrows = reinterpret_cast<const char*>(g_zero_rows6);
  const int nk = (a.dbg & 2) ? 2 : a.K / BK;  // dbg 2: two K tiles only (isolates the per-tile overhead in A/B runs)

  // ---- LDS-DMA descriptors: lane -> (row r8 of an 8-row piece, 16-byte slot); the slot holds source chunk
  // slot ^ ((LDS row >> 1) & 7).  A: wave w stages LDS rows [32 w, +32) as pieces 0..3; W: piece w of each region.
  // Round 6 (gemm8p.hip): buffer descriptors - SGPR base + 32-bit lane offset (0x80000000 = outside M / N: the hardware returns
  // zeros) + the K offset in an SGPR; no VALU in the DMA issue
  unsigned a_off[4], w_off[3];
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)((((long)a.M - 1) * a.lda + a.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)((((long)a.N - 1) * a.ldw + a.K) * 2), 0x00020000);
#define R6_BL16(RS, VOFF, SOFF, DST) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(DST), 16, (int)(VOFF), (int)(SOFF), 0, 0)
#define R6_TILE_SETUP(TMI, TNI)                                                                               \
  {                                                                                                           \
    const int d_m0 = (TMI) * BM, d_n0 = (TNI) * BN;                                                           \
    int ln_ = lane; /* opaque: the descriptors are computed here, per tile, not hoisted and carried (gemm8p.hip) */ \
    asm volatile("" : "+v"(ln_));                                                                             \
    const int r8 = ln_ >> 3, slot = ln_ & 7;                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
      const int i = 32 * wave + 8 * j + r8;                                                                   \
      const int chunk = slot ^ ((i >> 1) & 7);                                                                \
      const int gm = d_m0 + i; /* M < 2^31 - 512 (dispatcher) */                                              \
      a_off[j] = gm < a.M ? (unsigned)(((long)gm * a.lda + chunk * 8) * 2) : 0x80000000u;                     \
    }                                                                                                         \
    _Pragma("unroll") for (int nt = 0; nt < 3; ++nt) {                                                        \
      const int q = 8 * wave + r8; /* row inside the 64-row region: column half q >> 5, column q & 31 of block nt */ \
      const int chunk = slot ^ ((q >> 1) & 7);                                                                \
      const int gn = d_n0 + 96 * (q >> 5) + 32 * nt + (q & 31);                                               \
      w_off[nt] = gn < a.N ? (unsigned)(((long)gn * a.ldw + chunk * 8) * 2) : 0x80000000u;                    \
    }                                                                                                         \
  }
  // A piece J / W region NT at K position KP (in K tiles) into LDS buffer BSEL
#define R6_ISSUE_A(J, KP, BSEL) R6_BL16(rs_a, a_off[J], (KP) * (BK * 2), smem + (BSEL) * BUF + (4 * wave + (J)) * 1024);
#define R6_ISSUE_W(NT_, KP, BSEL) \
  R6_BL16(rs_w, w_off[NT_], (KP) * (BK * 2), smem + (BSEL) * BUF + TILE_A + ((NT_) * 8 + wave) * 1024);

  // ---- fragment read addresses: row = block_row0 + l31 (block_row0 % 32 == 0), 16-byte slot (2g + h) ^ ((l31 >> 1) & 7)
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const int sw = (l31 >> 1) & 7;
  unsigned rd[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) rd[g] = (unsigned)(l31 * ROWB + (((2 * g + h) ^ sw) << 4));
  const unsigned a_row0 = (unsigned)((128 * wr + 64 * wm) * ROWB);  // + mt * 32 rows
  const unsigned w_row0 = (unsigned)(TILE_A + (32 * wc) * ROWB);    // + nt * 64 rows

  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 af[2][4], bf0[4], bf1[4];  // A fragments [mt][k-group] of the K tile; W fragments of the current / next block

#define R6_READ_A(SB)                                                                                         \
  _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int g = 0; g < 4; ++g)               \
      R6_DS_READ(af[mt][g], (SB) + a_row0 + rd[g], (mt * 32) * ROWB);
#define R6_READ_W(BF, NT_, SB) \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) R6_DS_READ(BF[g], (SB) + w_row0 + rd[g], ((NT_) * 64) * ROWB);
  // waits: every register the covered reads write is a read-write operand, so no consumer can move above the wait
#define R6_WAIT_LGKM_W(BF) \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(BF[0]), "+v"(BF[1]), "+v"(BF[2]), "+v"(BF[3])::"memory")
#define R6_WAIT_LGKM_AW(BF)                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
               : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]), "+v"(af[1][1]), \
                 "+v"(af[1][2]), "+v"(af[1][3]), "+v"(BF[0]), "+v"(BF[1]), "+v"(BF[2]), "+v"(BF[3])::"memory")
  // D[n][m] += W[n][k] A[m][k]: the W fragment is the first operand, so a lane owns 4 consecutive n of one m
#define R6_MFMA(NT_, BF)                                                                                       \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                 \
      acc[NT_][mt] = mfma_h16_32x32x16(BF[g],              \
                                                             af[mt][g], acc[NT_][mt]);
#define R6_PHASE(WAIT, MF)                       \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
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

  // tile coordinates advance incrementally (one scalar division pair here, none per tile)
  int c_tm = (int)(((long)xcd * per_xcd + li) / NT), c_tn = (int)(((long)xcd * per_xcd + li) % NT);
  const int step_m = (int)(wg_per_xcd / NT), step_n = (int)(wg_per_xcd % NT);

  // ---- prologue = everything the steady-state stream has issued before P1 of its position 0: all of K tile 0 and
  // P3(-1)'s share of K tile 1; K tile 0 complete (nk >= 2 is guaranteed by the dispatcher)
  R6_TILE_SETUP(c_tm, c_tn)
  R6_ISSUE_A(0, 0, 0) R6_ISSUE_A(1, 0, 0) R6_ISSUE_W(0, 0, 0) R6_ISSUE_A(2, 0, 0) R6_ISSUE_A(3, 0, 0)
  R6_ISSUE_W(1, 0, 0) R6_ISSUE_W(2, 0, 0)
  R6_ISSUE_A(0, 1, 1) R6_ISSUE_A(1, 1, 1) R6_ISSUE_W(0, 1, 1)
  R6_WAIT_VM(3);
  __builtin_amdgcn_s_barrier();
  if (stagger && wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0 from here on

  unsigned gk = 0;  // stream position of the math (parity = LDS buffer)
  for (;;) {
    const long m0 = (long)c_tm * BM;
    const int n0 = c_tn * BN;
    const long li_next = li + wg_per_xcd;
    const bool has_next = li_next < per_xcd && (long)xcd * per_xcd + li_next < nblk;
    int n_tm = c_tm + step_m, n_tn = c_tn + step_n;
    if (n_tn >= NT) {
      n_tn -= NT;
      ++n_tm;
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // the descriptors describe the math's own tile at kt == 0 (the stream moved on to it at kt == nk - 2 of the previous
    // tile); rebuilt rather than carried across the epilogue (gemm8p.hip)
    if (gk != 0) R6_TILE_SETUP(c_tm, c_tn)

    // Round 6: K tiles with kt + 2 < nk run the STEADY copy of the body, the last two of every output tile the general one
    int kt = 0;
    for (; kt + 2 < nk; ++kt, ++gk) {
#define R6_STEADY 1
#include "gemm6p_ktile.inc"
#undef R6_STEADY
    }
    for (; kt < nk; ++kt, ++gk) {
#define R6_STEADY 0
#include "gemm6p_ktile.inc"
#undef R6_STEADY
    }
    li = li_next;
    c_tm = n_tm;
    c_tn = n_tn;

    // ---------------------------------------------------------------- epilogue (staged row writers, gemm_device.h)
    if (!(a.dbg & 256)) {  // dbg 256: no epilogue at all (tuning experiments)
      char* ws = smem + 2 * BUF + wave * SLICE;
      const long mw0 = m0 + 128 * wr + 64 * wm;
      const int nw0 = n0 + 96 * wc;
      int lane_e = lane;  // opaque copy: the epilogue's lane-dependent addresses are built here, not before the K loop
      asm volatile("" : "+v"(lane_e));
      const bool full_tile = m0 + BM <= a.M && n0 + BN <= a.N;
      if constexpr (sizeof(TOUT) == 2) {
        bf16_t* Cbb = reinterpret_cast<bf16_t*>(a.C);
        if (full_tile) epi_staged_bf16<TM, TN, ACT, true>(acc, a, Cbb, ws, mw0, nw0, lane_e);
        else epi_staged_bf16<TM, TN, ACT, false>(acc, a, Cbb, ws, mw0, nw0, lane_e);
      } else {
        float* Cbf = reinterpret_cast<float*>(a.C);
        if (full_tile) epi_staged_f32<TM, TN, ACT, true>(acc, a, Cbf, a.res, ws, mw0, nw0, lane_e);
        else epi_staged_f32<TM, TN, ACT, false>(acc, a, Cbf, a.res, ws, mw0, nw0, lane_e);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), visible to the compiler: see gemm8p.hip (f32 row writer)
      }
    }
    if (!has_next) break;
  }
  if (stagger && wr == 0) __builtin_amdgcn_s_barrier();  // group 0 meets group 1's extra barrier
#undef R6_PHASE
#undef R6_MFMA
#undef R6_WAIT_LGKM_AW
#undef R6_WAIT_LGKM_W
#undef R6_READ_W
#undef R6_READ_A
#undef R6_ISSUE_W
#undef R6_ISSUE_A
#undef R6_TILE_SETUP
}

template <typename TOUT, int ACT>
static int launch6p(const GemmArgs& a, hipStream_t stream, const char* epi_name) {
  constexpr int BM = 256, BN = 192;
  const long nblk = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const size_t lds = (size_t)2 * (BM + BN) * ROWB + 8 * 6144;  // 160 KiB: one persistent workgroup per CU
  const long gx = std::min<long>(((nblk + 7) / 8) * 8, 256);
  char pname[96];
  snprintf(pname, sizeof pname, "gemm6p_kernel<" ROMA_H16_NAME ",%s,dense,%s>", sizeof(TOUT) == 4 ? "f32" : ROMA_H16_NAME, epi_name);
  ProfScope ps(pname, 2.0 * (double)a.M * (a.n_alg > 0 ? a.n_alg : a.N) * (a.k_alg > 0 ? a.k_alg : a.K), "flop", stream);  // un-padded channels
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm6p_kernel<TOUT, ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm6p_kernel<TOUT, ACT>), dim3((unsigned)gx), dim3(512), lds, stream, a);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// 0 = launched, 1 = not this kernel's problem, < 0 = error.  Called by gemm8p_try_launch for the shapes gemm.hip would
// run on 256 x 192 tiles.
int gemm6p_try_launch(const GemmArgs& a, hipStream_t stream) {
  if (a.conv_c > 0 || a.mode != EPI_STD || a.res_bf16 || a.scale || a.qkv_pad) return 1;
  if (a.act == ACT_GELU) return 1;
  if ((reinterpret_cast<uintptr_t>(a.C) & 15) != 0) return 1;
  if (a.out_dt == DT_BF16) {
    if (a.res != nullptr || (a.ldc & 7) != 0) return 1;
    if (a.act == ACT_RELU) return launch6p<bf16_t, ACT_RELU>(a, stream, "relu");
    return launch6p<bf16_t, ACT_NONE>(a, stream, "none");
  }
  if (a.out_dt == DT_F32) {
    if ((a.ldc & 3) != 0 || (a.N & 3) != 0) return 1;
    if (a.res && ((a.ldr & 3) != 0 || (reinterpret_cast<uintptr_t>(a.res) & 15) != 0)) return 1;
    if (a.act == ACT_RELU) return launch6p<float, ACT_RELU>(a, stream, "relu");
    return launch6p<float, ACT_NONE>(a, stream, "none");
  }
  return 1;
}

}  // namespace roma
