// Four-wave bf16 MFMA GEMM for gfx950 (experimental sibling of gemm8p.hip): C[M,N] = epi(A[M,K] . W[N,K]^T), 256 x 256 x 32 tiles,
// ONE wave per SIMD with a 128 x 128 wave tile (256 accumulator registers of the 512 a single wave per SIMD may use).
//
// Why: the 8-phase kernels run the K loop at ~50 % of the MFMA peak.  Their phase time is 2 x max(load block, MFMA block),
// and the load block of a wave group (12 / 4 / 8 fragment reads + 2 LDS-DMA issues at ~100 cycles each) is ~1.8 x the
// 256-cycle MFMA block of its partner: the matrix pipe waits for the other group's issue slots.  Here every wave feeds
// its own matrix pipe: the fragment reads of the NEXT k-group and this wave's share of the LDS-DMA are issued BETWEEN the
// 16 MFMAs of the current k-group (an MFMA occupies the pipe for 32 cycles during which the wave is free to issue), the
// wave tile is twice as large (half the LDS bytes per FLOP: 0.5 KB / MFMA instead of 0.75) and there is one barrier per
// K tile instead of eight.
//
//   * wave (wr, wc) of 2 x 2 owns rows [128 wr, +128) x cols [128 wc, +128): TM 4 x TN 4 MFMA blocks (32x32x16).
//   * K tile = 32 (two k-groups): a stage is A 256 x 64 B + W 256 x 64 B = 32 KiB, four stages in a ring (128 KiB) + 4 x 8 KiB
//     epilogue slices = 160 KiB.  With K tiles of ~1 000 MFMA cycles the ring gives the DMA two full K tiles (~1 us)
//     between issue and first use.
//   * 64-byte LDS rows, 4 slots of 16 B; slot = chunk ^ ((row >> 2) & 3): the 16 lanes of a ds_read_b128 group sit on 4 x 4
//     rows with equal (row & 3) and distinct (row >> 2) & 3 -> all 64 banks, conflict free (rows 0-3, 12-15, 20-27 ...).
//   * per K tile T (one s_barrier): wait vmcnt(8) [tile T+1 landed, T+2 may fly], barrier [tile T-1's stage is dead for
//     everybody, T+1 visible], k-group 0: MFMAs of (T, 0) with the reads of (T, 1) and 4 DMA pieces of tile T+3
//     interleaved; lgkmcnt(0); k-group 1: MFMAs of (T, 1) with the reads of (T+1, 0) and the other 4 pieces; lgkmcnt(0).
//     The DMA is one stream over all (output tile, k) positions of the persistent workgroup, like gemm8p.
//   * epilogue: the staged row writers of gemm_device.h with TM = TN = 4 (256-byte staged rows).
#include "gemm.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gemm_device.h"
#include "gemm4w_acc.inc"

namespace roma {

static __device__ __attribute__((aligned(256))) unsigned int g_zero_rows4[16384];  // zero source for out-of-range rows

#define R4_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define R4_DS_READ(REG, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(REG) : "v"(ADDR), "n"(OFF))

template <typename TOUT, int ACT>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const GemmArgs a) {
  constexpr int BM = 256, BN = 256, BK = 32, RB4 = 64;  // RB4: bytes per LDS row
  constexpr int TILE_A = BM * RB4, STAGE = TILE_A + BN * RB4;  // 32 KiB
  constexpr int NST = 4;
  constexpr int TM = 4, TN = 4;
  constexpr int SLICE = 32 * TN * 64;  // 8 KiB epilogue staging per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- persistent tile walk, XCD-aware (as gemm8p)
  const int NT = (a.N + BN - 1) / BN;
  const long nblk = (long)((a.M + BM - 1) / BM) * NT;
  const long per_xcd = (nblk + 7) / 8;
  const int xcd = blockIdx.x % 8;
  const long wg_per_xcd = gridDim.x / 8;
  long li = blockIdx.x / 8;
  if (li >= per_xcd || (long)xcd * per_xcd + li >= nblk) return;

  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* Wb = reinterpret_cast<const bf16_t*>(a.W);
  const char* zrows = reinterpret_cast<const char*>(g_zero_rows4);
  const int nk = a.K / BK;

  // ---- LDS-DMA descriptors: wave w stages LDS rows [64 w, +64) of A and of W as 4 pieces of 16 rows each;
  // lane -> (row r = lane >> 2 of the piece, slot = lane & 3) holding source chunk slot ^ ((LDS row >> 2) & 3)
  const char* a_src[4];
  const char* w_src[4];
#define R4_TILE_SETUP(TMI, TNI)                                                                                \
  {                                                                                                            \
    const int d_m0 = (TMI) * BM, d_n0 = (TNI) * BN;                                                            \
    const int r = lane >> 2, slot = lane & 3;                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                            \
      const int i = 64 * wave + 16 * j + r;                                                                    \
      const int chunk = slot ^ ((i >> 2) & 3);                                                                 \
      const int gm = d_m0 + i, gn = d_n0 + i;                                                                  \
      a_src[j] = gm < a.M ? reinterpret_cast<const char*>(Ab + (long)gm * a.lda + chunk * 8) : zrows + chunk * 16; \
      w_src[j] = gn < a.N ? reinterpret_cast<const char*>(Wb + (long)gn * a.ldw + chunk * 8) : zrows + chunk * 16; \
    }                                                                                                          \
  }
#define R4_ISSUE_A(J, KP, ST) glds16(a_src[J] + (long)(KP) * (BK * 2), smem + (ST) * STAGE + (4 * wave + (J)) * 1024);
#define R4_ISSUE_W(J, KP, ST) glds16(w_src[J] + (long)(KP) * (BK * 2), smem + (ST) * STAGE + TILE_A + (4 * wave + (J)) * 1024);

  // ---- fragment read addresses: row = block_row0 + l31 (block_row0 % 32 == 0), slot (2g + h) ^ ((l31 >> 2) & 3)
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const int sw = (l31 >> 2) & 3;
  unsigned rd[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) rd[g] = (unsigned)(l31 * RB4 + (((2 * g + h) ^ sw) << 4));
  const unsigned a_row0 = (unsigned)((128 * wr) * RB4);           // + mt * 32 rows
  const unsigned w_row0 = (unsigned)(TILE_A + (128 * wc) * RB4);  // + nt * 32 rows

  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 fa[2][4], fw[2][4];  // [buffer][block]

#define R4_READ_A(BUFI, MT, SB, G) R4_DS_READ(fa[BUFI][MT], (SB) + a_row0 + rd[G], (MT) * 32 * RB4);
#define R4_READ_W(BUFI, NT_, SB, G) R4_DS_READ(fw[BUFI][NT_], (SB) + w_row0 + rd[G], (NT_) * 32 * RB4);
#define R4_WAIT_LGKM(BUFI)                                                                                      \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                           \
               : "+v"(fa[BUFI][0]), "+v"(fa[BUFI][1]), "+v"(fa[BUFI][2]), "+v"(fa[BUFI][3]), "+v"(fw[BUFI][0]),   \
                 "+v"(fw[BUFI][1]), "+v"(fw[BUFI][2]), "+v"(fw[BUFI][3])::"memory")
  // Accumulators in HARD AGPRs: tile T = tn * 4 + tm is a[16 T : 16 T + 15] (gemm4w_acc.inc).  hipcc does not keep 256
  // accumulator registers resident across the loop - as a builtin, or as asm with a "+a" operand, it held them in VGPRs /
  // scratch and copied 16 registers into AGPRs in front of EVERY MFMA (324 v_accvgpr_write + 183 scratch loads per K
  // tile).  Every asm lists the range it writes as clobbered; the arch-VGPR pressure stays far below 256, so the compiler
  // never spills into AGPRs (tools/audit_gemm8p_isa.py: no v_accvgpr instruction outside the asm blocks).  An
  // accumulator is touched once per k-group (16 MFMAs apart): no MFMA -> MFMA hazard nops inside the loop; the reads
  // after the loop are fenced by hand (R4_ACC_FENCE).
#define R4_MFMA(BUFI, NT_, MT, T)                                                                              \
  asm volatile("v_mfma_f32_32x32x16_bf16 " R4_ARANGE_##T ", %0, %1, " R4_ARANGE_##T ::"v"(fw[BUFI][NT_]), "v"(fa[BUFI][MT]) \
               : R4_ACLOB_##T);
#define R4_ACC_FENCE asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")
#define R4_SB __builtin_amdgcn_sched_barrier(0);
  // one k-group: 16 MFMAs on buffer CUR, with the 8 fragment reads of the next k-group (buffer NXT, stage base SBN, k-group
  // GN) and 4 LDS-DMA pieces (ISS0..ISS3, possibly empty) in the gaps
#define R4_KGROUP(CUR, NXT, SBN, GN, ISS0, ISS1, ISS2, ISS3)                                     \
  R4_MFMA(CUR, 0, 0, 0) R4_SB R4_READ_A(NXT, 0, SBN, GN) R4_SB                                  \
  R4_MFMA(CUR, 0, 1, 1) R4_SB R4_READ_A(NXT, 1, SBN, GN) R4_SB                                  \
  R4_MFMA(CUR, 0, 2, 2) R4_SB R4_READ_A(NXT, 2, SBN, GN) R4_SB                                  \
  R4_MFMA(CUR, 0, 3, 3) R4_SB R4_READ_A(NXT, 3, SBN, GN) R4_SB                                  \
  R4_MFMA(CUR, 1, 0, 4) R4_SB R4_READ_W(NXT, 0, SBN, GN) R4_SB                                  \
  R4_MFMA(CUR, 1, 1, 5) R4_SB R4_READ_W(NXT, 1, SBN, GN) R4_SB                                  \
  R4_MFMA(CUR, 1, 2, 6) R4_SB R4_READ_W(NXT, 2, SBN, GN) R4_SB                                  \
  R4_MFMA(CUR, 1, 3, 7) R4_SB R4_READ_W(NXT, 3, SBN, GN) R4_SB                                  \
  R4_MFMA(CUR, 2, 0, 8) R4_SB ISS0 R4_SB                                                        \
  R4_MFMA(CUR, 2, 1, 9) R4_SB ISS1 R4_SB                                                        \
  R4_MFMA(CUR, 2, 2, 10) R4_SB ISS2 R4_SB                                                       \
  R4_MFMA(CUR, 2, 3, 11) R4_SB ISS3 R4_SB                                                       \
  R4_MFMA(CUR, 3, 0, 12) R4_MFMA(CUR, 3, 1, 13) R4_MFMA(CUR, 3, 2, 14) R4_MFMA(CUR, 3, 3, 15) R4_SB

  int c_tm = (int)(((long)xcd * per_xcd + li) / NT), c_tn = (int)(((long)xcd * per_xcd + li) % NT);
  const int step_m = (int)(wg_per_xcd / NT), step_n = (int)(wg_per_xcd % NT);

  // ---- prologue: K tiles 0, 1, 2 of the first output tile under way (nk >= 4 guaranteed), tile 0 landed and visible,
  // fragments of (0, k-group 0) in buffer 0
  R4_TILE_SETUP(c_tm, c_tn)
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    R4_ISSUE_A(0, t, t) R4_ISSUE_A(1, t, t) R4_ISSUE_A(2, t, t) R4_ISSUE_A(3, t, t)
    R4_ISSUE_W(0, t, t) R4_ISSUE_W(1, t, t) R4_ISSUE_W(2, t, t) R4_ISSUE_W(3, t, t)
  }
  R4_WAIT_VM(16);
  __builtin_amdgcn_s_barrier();
  {
    const unsigned sb0 = lds0;
    R4_READ_A(0, 0, sb0, 0) R4_READ_A(0, 1, sb0, 0) R4_READ_A(0, 2, sb0, 0) R4_READ_A(0, 3, sb0, 0)
    R4_READ_W(0, 0, sb0, 0) R4_READ_W(0, 1, sb0, 0) R4_READ_W(0, 2, sb0, 0) R4_READ_W(0, 3, sb0, 0)
    R4_WAIT_LGKM(0);
  }

  unsigned gt = 0;  // stream position of the math in K tiles (stage = gt & 3)
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
    R4_ACC_ZERO

    for (int kt = 0; kt < nk; ++kt, ++gt) {
      const unsigned st = gt & 3u;
      const unsigned sb = lds0 + st * STAGE, sbn = lds0 + ((gt + 1u) & 3u) * STAGE;
      // stream positions: +1 (its k-group 0 is read at the end of this tile), +2 (in flight), +3 (issued during this tile)
      const bool e2 = kt + 2 < nk || has_next;
      const bool e3 = kt + 3 < nk || has_next;
      const int k3 = kt + 3 < nk ? kt + 3 : kt + 3 - nk;
      const unsigned st3 = (gt + 3u) & 3u;
      if (kt + 3 == nk && has_next) R4_TILE_SETUP(n_tm, n_tn)  // the DMA stream enters the next output tile
      if (e2) {
        R4_WAIT_VM(8);
      } else {
        R4_WAIT_VM(0);
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (e3) {
        R4_KGROUP(0, 1, sb, 1, R4_ISSUE_A(0, k3, st3), R4_ISSUE_A(1, k3, st3), R4_ISSUE_A(2, k3, st3), R4_ISSUE_A(3, k3, st3))
        R4_WAIT_LGKM(1);
        R4_SB
        R4_KGROUP(1, 0, sbn, 0, R4_ISSUE_W(0, k3, st3), R4_ISSUE_W(1, k3, st3), R4_ISSUE_W(2, k3, st3), R4_ISSUE_W(3, k3, st3))
        R4_WAIT_LGKM(0);
        R4_SB
      } else {
        R4_KGROUP(0, 1, sb, 1, , , , )
        R4_WAIT_LGKM(1);
        R4_SB
        R4_KGROUP(1, 0, sbn, 0, , , , )
        R4_WAIT_LGKM(0);
        R4_SB
      }
    }
    li = li_next;
    c_tm = n_tm;
    c_tn = n_tn;
    R4_ACC_FENCE;  // the last MFMAs (asm: invisible to the hazard recogniser) retire before the accumulators are read

    // ---------------------------------------------------------------- epilogue (staged row writers, gemm_device.h)
    // one 32-row block of the wave tile at a time: its 4 accumulator tiles (64 registers) come out of the AGPRs, then the
    // TM = 1 row writer runs on them
    if (!(a.dbg & 256)) {
      char* ws = smem + NST * STAGE + wave * SLICE;
      const int nw0 = n0 + 128 * wc;
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const bool full_tile = m0 + BM <= a.M && n0 + BN <= a.N;
      bf16_t* Cbb = reinterpret_cast<bf16_t*>(a.C);
#define R4_EPI_TM(TMI, T0, T1, T2, T3)                                                                         \
  {                                                                                                            \
    f32x16 accr[TN][1];                                                                                        \
    R4_ACC_READ_##T0(accr[0][0]) R4_ACC_READ_##T1(accr[1][0]) R4_ACC_READ_##T2(accr[2][0]) R4_ACC_READ_##T3(accr[3][0]) \
    const long mw0 = m0 + 128 * wr + 32 * (TMI);                                                               \
    if (full_tile) epi_staged_bf16<1, TN, ACT, true>(accr, a, Cbb, ws, mw0, nw0, lane_e);                      \
    else epi_staged_bf16<1, TN, ACT, false>(accr, a, Cbb, ws, mw0, nw0, lane_e);                               \
  }
      R4_EPI_TM(0, 0, 4, 8, 12)
      R4_EPI_TM(1, 1, 5, 9, 13)
      R4_EPI_TM(2, 2, 6, 10, 14)
      R4_EPI_TM(3, 3, 7, 11, 15)
#undef R4_EPI_TM
    }
    if (!has_next) break;
  }
  R4_WAIT_VM(0);
#undef R4_KGROUP
#undef R4_SB
#undef R4_MFMA
#undef R4_ACC_FENCE
#undef R4_WAIT_LGKM
#undef R4_READ_W
#undef R4_READ_A
#undef R4_ISSUE_W
#undef R4_ISSUE_A
#undef R4_TILE_SETUP
}

template <typename TOUT, int ACT>
static int launch4w(const GemmArgs& a, hipStream_t stream, const char* epi_name) {
  constexpr int BM = 256, BN = 256;
  const long nblk = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const size_t lds = (size_t)4 * (BM + BN) * 64 + 4 * 8192;  // 160 KiB: one persistent workgroup per CU
  const long gx = std::min<long>(((nblk + 7) / 8) * 8, 256);
  char pname[96];
  snprintf(pname, sizeof pname, "gemm4w_kernel<bf16,bf16,dense,%s>", epi_name);
  ProfScope ps(pname, 2.0 * (double)a.M * a.N * a.K, "flop", stream);
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<TOUT, ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm4w_kernel<TOUT, ACT>), dim3((unsigned)gx), dim3(256), lds, stream, a);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// 0 = launched, 1 = not this kernel's problem.  Plain bf16 problems only (bias + none / ReLU / GELU); called by
// gemm8p_try_launch when the "gemm4w" tuning switch is on.
int gemm4w_try_launch(const GemmArgs& a, hipStream_t stream) {
  if (a.conv_c > 0 || a.mode != EPI_STD || a.res_bf16 || a.scale || a.qkv_pad || a.res) return 1;
  if (a.out_dt != DT_BF16 || (a.ldc & 7) != 0 || (reinterpret_cast<uintptr_t>(a.C) & 15) != 0) return 1;
  if (a.K % 32 != 0 || a.K < 128) return 1;
  if (a.act == ACT_GELU) return launch4w<bf16_t, ACT_GELU>(a, stream, "gelu");
  if (a.act == ACT_RELU) return launch4w<bf16_t, ACT_RELU>(a, stream, "relu");
  return launch4w<bf16_t, ACT_NONE>(a, stream, "none");
}

}  // namespace roma
