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

// the five instantiation families of gemm_kernel (gemm_kernel.inc), one translation unit each
int gemm_family_f32(const GemmArgs& a, hipStream_t stream);
int gemm_family_f32_conv(const GemmArgs& a, hipStream_t stream);
int gemm_family_h16(const GemmArgs& a, hipStream_t stream);
int gemm_family_h16_conv(const GemmArgs& a, hipStream_t stream);
int gemm_family_h16f32(const GemmArgs& a, hipStream_t stream);

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
    ROMA_REQUIRE(a.conv_korder == 0 || (a.conv_korder == 1 && a.conv_c % 64 == 0), "gemm(conv3x3): slab-major weights need Cin % 64 == 0");
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
  if (a.in_dt == DT_F32 && a.out_dt == DT_F32) return conv ? gemm_family_f32_conv(a, stream) : gemm_family_f32(a, stream);
  if (a.in_dt == DT_BF16 && a.out_dt == DT_BF16) return conv ? gemm_family_h16_conv(a, stream) : gemm_family_h16(a, stream);
  if (a.in_dt == DT_BF16 && a.out_dt == DT_F32 && !conv) return gemm_family_h16f32(a, stream);
  set_error("gemm: unsupported dtype combination (3x3 convolutions keep their input type)");
  return -1;
}

}  // namespace roma
