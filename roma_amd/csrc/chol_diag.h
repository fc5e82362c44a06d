// Cholesky factorisation + explicit inverse of ONE 64 x 64 diagonal block held in LDS, as two cooperating waves.
// Shared by chol_diag_kernel (elementwise.hip: the right-looking chain of rounds 1-5, still used for separate right-hand
// sides) and chol_col_kernel (chol_col.hip, round 6: the left-looking block-column kernel of the GP's augmented system).
//
//   L  [64][CHOL_S]   on entry the SPD block (row i = lane i), on exit the factor in its lower triangle (the strictly upper
//                     part holds harmless intermediate values: every consumer masks it)
//   LT [64][CHOL_S]   LT[j][c] = L[c][j]: column j of the factor as one contiguous row, the broadcast source of the updates
//   XT [64][CHOL_S]   XT[c][t] = X[t][c], X = L^-1 (lane c solves L x = e_c)
//   progress          last finished column step of the factorisation (workgroup-scope release / acquire)
//
// 68-float rows: 16-byte aligned and conflict free for lane-per-row ds_read_b128 (bank = 4 * lane).  The inner loops move 4
// columns per LDS operation (row piece of this lane + a broadcast piece of the transposed copy): the scalar version issued
// ~10k dependent single-dword LDS operations per call and took 123 us on the GP's critical path.
#pragma once
#include "common.h"

namespace roma {

constexpr int CHOL_S = 68;

// wave A (lane i owns row i): the factorisation, publishing every finished column in `progress`.
// Round 6: blocked by bands of 16 columns.  Inside a band the lane's 16 entries of its row live in REGISTERS and the column
// being eliminated is broadcast with v_readlane (the pivot and the 15 - jj entries l_cj of the band's later rows are lane
// constants: the 64 steps are fully unrolled), so a column step is ~sqrt + divide + (15 - jj) readlane / fma pairs with no LDS
// round trip on its dependency chain; the columns right of the band receive the band's 16 rank-1 updates at once, from
// LDS broadcasts, when the band is done.  Until round 5 every step updated ALL later columns through LDS (2 reads + 1 write
// of 16 bytes per 4 columns, and the next pivot waited for them): 36 us per block, 0.9 ms of the GP chain.  Same operations
// on every element; the updates of one element are applied in the same order j = 0, 1, 2 ... (bit-identical factor).
__device__ __forceinline__ void chol_diag_factor_wave(float* L, float* LT, int* progress, int i) {
  constexpr int S = CHOL_S;
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    float a[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&L[i * S + 16 * jb + 4 * q]);
      a[4 * q] = v[0]; a[4 * q + 1] = v[1]; a[4 * q + 2] = v[2]; a[4 * q + 3] = v[3];
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = 16 * jb + jj;
      const float ajj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[jj]), j));
      const float d = sqrtf(ajj);
      const float lij = (i == j) ? d : a[jj] / d;
      a[jj] = lij;
      L[i * S + j] = lij;   // rows < j: harmless values in the strictly upper part (every consumer masks it)
      LT[j * S + i] = lij;  // column j of the factor, contiguous: the broadcast source of the band update below
      // columns up to j are final: let the inverse take the rows up to j.  Release / acquire at workgroup scope (round 5,
      // ADVICE r04).  Published every fourth column: the release waits for this wave's LDS writes to land (~100 cycles on the
      // factorisation's critical path), and the inverse consumes rows four at a time anyway.
      if ((jj & 3) == 3) __hip_atomic_store(progress, j, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int cc = jj + 1; cc < 16; ++cc) {
        const float lcj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lij), 16 * jb + cc));
        a[cc] = a[cc] - lij * lcj;
      }
    }
    // the columns right of the band: a_ic -= l_ij l_cj for the band's 16 columns j, in the order of j
    if (jb < 3) {
#pragma unroll
      for (int c = 16 * (jb + 1); c < 64; c += 4) {
        f32x4 t = *reinterpret_cast<const f32x4*>(&L[i * S + c]);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const f32x4 lc = *reinterpret_cast<const f32x4*>(&LT[(16 * jb + jj) * S + c]);  // same address in every lane: broadcast
#pragma unroll
          for (int u = 0; u < 4; ++u) t[u] = t[u] - a[jj] * lc[u];
        }
        *reinterpret_cast<f32x4*>(&L[i * S + c]) = t;
      }
    }
  }
}

// wave B (lane c owns unknown vector c): L x = e_c by forward substitution, a few rows behind the factorisation.  Four
// interleaved partial sums (t mod 4) keep the dependent-FMA chain at 16 instead of 64 per row.  Row r reads L[r][t <= r] only:
// final once the factorisation has published column r.
// Round 6: the unknowns live in REGISTERS (64 per lane, rows fully unrolled) - until round 5 every x_r went to LDS and came
// back for the next row (a write -> read round trip of ~200 cycles on the chain of all 64 rows), which made this wave, not the
// factorisation, the long pole of the block (31 of 36 us).  The L rows are broadcast reads that depend on nothing but the
// progress flag, so they are in flight rows ahead.  Same terms, same four partial sums, same order: bit-identical inverse.
__device__ __forceinline__ void chol_diag_inverse_wave(const float* L, float* XT, int* progress, int c) {
  constexpr int S = CHOL_S;
  float x[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    if ((r & 3) == 0)  // rows r .. r + 3 are published together (chol_diag_factor_wave)
      while (__hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < r + 3) __builtin_amdgcn_s_sleep(1);
    float part[4] = {(r == c) ? 1.f : 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < r; t += 4) {
      const f32x4 lr = *reinterpret_cast<const f32x4*>(&L[r * S + t]);  // broadcast
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (t + u < r) part[u] = part[u] - lr[u] * x[t + u];
    }
    x[r] = ((part[0] + part[1]) + (part[2] + part[3])) / L[r * S + r];
    if ((r & 3) == 3) *reinterpret_cast<f32x4*>(&XT[c * S + r - 3]) = f32x4{x[r - 3], x[r - 2], x[r - 1], x[r]};
  }
}

// write-back by `nthreads` threads, 16 bytes per lane: the factor (upper part zeroed) into the matrix, Linv^T rows (= XT
// rows) and Linv (transposed read) into the per-block inverse tables
__device__ __forceinline__ void chol_diag_writeback(const float* L, const float* XT, float* Ab, long ld, float* Li, float* LiT,
                                                    int tid, int nthreads) {
  constexpr int S = CHOL_S;
  for (int idx = tid; idx < 64 * 16; idx += nthreads) {
    const int row = idx >> 4, c4 = (idx & 15) * 4;
    f32x4 lv = *reinterpret_cast<const f32x4*>(&L[row * S + c4]);
    f32x4 xt = *reinterpret_cast<const f32x4*>(&XT[row * S + c4]);  // XT[row][c4..] = X[c4..][row] = LinvT[row][c4..]
    f32x4 xl;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c4 + u > row) lv[u] = 0.f;
      if (c4 + u < row) xt[u] = 0.f;               // X[t][c] is zero for t < c
      xl[u] = (c4 + u <= row) ? XT[(c4 + u) * S + row] : 0.f;  // Linv[row][c4+u] = X[row][c4+u]
    }
    if (Ab) *reinterpret_cast<f32x4*>(Ab + (long)row * ld + c4) = lv;  // (chol_col.hip keeps the factor block out of A)
    *reinterpret_cast<f32x4*>(LiT + row * 64 + c4) = xt;
    *reinterpret_cast<f32x4*>(Li + row * 64 + c4) = xl;
  }
}

}  // namespace roma
