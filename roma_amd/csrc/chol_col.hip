// Left-looking block-column step of the GP's blocked Cholesky solve (matcher.py:291-323: L = cholesky(K_yy + sigma^2 I),
// cholesky_solve) on the AUGMENTED system: one (n + d) x n matrix per image, K_yy on top and the d right-hand-side rows F^T
// right behind it (model.hip gp_posterior), so that the forward substitution is part of the factorisation.
//
// Rounds 1-5 ran the right-looking form: per 64-wide step k one chol_diag launch (8 workgroups on the whole chip), one
// panel GEMM and one trailing-update GEMM (K = 64: output bound) - 75 dependent launches of 10 .. 110 us for n = 1600, 2.6 ms
// of a sub-batch stream's wall clock with both streams inside the chain at the same time (profiles/r05_final_stream_overlap.txt).
// Left-looking, ONE launch per block column k does everything that column needs:
//     T        = A[S, S]  - sum_{t < k} L[S, t] L[S, t]^T          S = rows [64 k, 64 k + 64)
//     L[S, S]  = chol(T),  Linv_kk = L[S, S]^-1
//     L[R, S]  = (A[R, S] - sum_{t < k} L[R, t] L[S, t]^T) Linv_kk^T     for every row block R below S, incl. the d rhs rows
// Per image one LEADER workgroup (block x = 0) computes T and factorises it (the ~36 us two-wave chol_diag pair: the critical
// path), publishes the inverse table of the block and raises a flag; the FOLLOWER workgroups (one per row block R) meanwhile
// form their product P = A[R, S] - sum_t ... (independent of the factorisation), then wait for the flag, multiply by Linv^T
// and store.  The hand-off is the agent-scope release / acquire recipe of cdna_hip_programming.md (Guideline 16): plain
// stores, __syncthreads, one lane: release fence + vmcnt(0) + relaxed flag store; consumer: ONE lane polls relaxed with
// s_sleep, one acquire fence, __syncthreads, plain loads.  The wait is BOUNDED: HIP promises nothing about dispatch order, so
// a follower whose leader does not answer within ~0.5 ms factorises its own copy of T (the redundant form below) - slower,
// never wrong, never a hang.  The flag carries (call epoch, k + 1), so no value left by an earlier call or column can match.
//   * first version of this file (profiles/r06_v1_gp_block_column_redundant_diag.log): EVERY workgroup recomputed T and its
//     factorisation - no hand-off at all.  Alone 2.62 -> 2.14 ms per solve, but with both sub-batch streams inside the chain
//     (the benchmark's regime) 3.0 -> 3.4 ms: 512 workgroups x 4 waves of redundant f32 MFMA work, two per CU.  That form is
//     what column 0 still runs (its T is A[S, S] itself - nothing to recompute) and what the time-out path falls back to.
// Inside a workgroup (4 waves): 32 x 32 tiles on v_mfma_f32_32x32x2_f32 (exact f32), operands straight from L2 into
// registers (lane (i, kk) holds row i, columns 32 kk + [0, 32) of a 64-deep slab: both operands use the same k pairing, so
// no LDS staging and no barrier in the K loop), slabs double buffered; the product with Linv^T takes its operands from LDS;
// stores go to L[R, S] and - transposed, 16 bytes per lane - to LT[S, R], which removes the separate n x n transpose launch
// in front of the backward substitution.  25 launches instead of 75 + 1 for n = 1600.  The summation order differs from the
// right-looking form (one fmaf chain over t inside the accumulator instead of k in-memory subtractions), so results agree
// to rounding, not bit for bit (tests/test_gpu_ops.py::test_cholesky_solve*).
#include "chol_diag.h"
#include "elementwise.h"

namespace roma {

namespace {

struct CFrag { f32x4 v[8]; };

__device__ __forceinline__ void cc_load(CFrag& f, const float* p) {
#pragma unroll
  for (int q = 0; q < 8; ++q) f.v[q] = *reinterpret_cast<const f32x4*>(p + 4 * q);
}
// D[i][j] += sum_k a(i, k) b(j, k) over the 64 k values of a slab (lane half kk = the second k of every MFMA's pair)
__device__ __forceinline__ f32x16 cc_mfma(const CFrag& a, const CFrag& b, f32x16 acc) {
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q][u], b.v[q][u], acc, 0, 0, 0);
  return acc;
}
// one 32 x 32 tile of  sum_{t < k} X[rows pa][slab t] . Y[rows pb][slab t]^T ;  pa / pb: this lane's row, column 32 kk
__device__ __forceinline__ f32x16 cc_tile_product(const float* pa, const float* pb, int k) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  CFrag a0, b0, a1, b1;
  if (k > 0) {
    cc_load(a0, pa);
    cc_load(b0, pb);
  }
  for (int t = 0; t < k; t += 2) {
    if (t + 1 < k) {
      cc_load(a1, pa + 64 * (t + 1));
      cc_load(b1, pb + 64 * (t + 1));
    }
    acc = cc_mfma(a0, b0, acc);
    if (t + 1 < k) {
      if (t + 2 < k) {
        cc_load(a0, pa + 64 * (t + 2));
        cc_load(b0, pb + 64 * (t + 2));
      }
      acc = cc_mfma(a1, b1, acc);
    }
  }
  return acc;
}

constexpr int CC_POLLS = 2000;  // x (relaxed L2 load + s_sleep 8): ~0.5 ms before a follower gives up on its leader

}  // namespace

__global__ __launch_bounds__(256, 2) void chol_col_kernel(float* __restrict__ A, long ld, long strideA, float* __restrict__ LTm,
                                                          long strideLT, int n, float* __restrict__ Linv,
                                                          float* __restrict__ LinvT, int k, int nblk, unsigned epoch,
                                                          int use_leader) {
  constexpr int S = CHOL_S;
  __shared__ __attribute__((aligned(16))) float L[64 * S];
  __shared__ __attribute__((aligned(16))) float LT[64 * S];
  __shared__ __attribute__((aligned(16))) float XT[64 * S];
  __shared__ __attribute__((aligned(16))) float Pb[64 * S];
  __shared__ int progress;
  __shared__ int leader_ok;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, kk = lane >> 5;
  const int img = blockIdx.y;
  // column 0 (and use_leader = 0): every workgroup owns a row block and factorises its own copy; otherwise block x = 0 is the
  // image's leader (no row block) and blocks x >= 1 are the followers
  const bool split = use_leader && k > 0;
  const bool leader = split && blockIdx.x == 0;
  const int rb = split ? (int)blockIdx.x - 1 : (int)blockIdx.x;
  float* Ai = A + (long)img * strideA;
  const long S0 = 64l * k, R0 = 64l * (k + 1) + 64l * rb;
  // the hand-off flag of this image lives in LT's first diagonal block, element [1][0] (strictly lower part of a block that
  // holds an UPPER triangular L^T: nobody reads it - chol_col_restore_kernel masks it)
  unsigned* flag = reinterpret_cast<unsigned*>(LTm + (long)img * strideLT + n);
  const unsigned want = (epoch << 8) | (unsigned)(k + 1);
  float* Li = Linv + ((long)img * nblk + k) * 4096;
  float* LiT = LinvT + ((long)img * nblk + k) * 4096;
  if (tid == 0) {
    progress = -1;
    leader_ok = 0;
  }
  const int ti = wave >> 1, tj = wave & 1;  // this wave's 32 x 32 tile of a 64 x 64 product

  // ---- followers (and the redundant form): P = A[R, S] - sum_t L[R, t] L[S, t]^T, one tile per wave -> LDS
  if (!leader) {
    const float* pa = Ai + (R0 + 32 * ti + l31) * ld + 32 * kk;
    const float* pb = Ai + (S0 + 32 * tj + l31) * ld + 32 * kk;
    const f32x16 acc = cc_tile_product(pa, pb, k);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kk, col = 32 * tj + l31;
      Pb[row * S + col] = Ai[(R0 + row) * ld + S0 + col] - acc[r];
    }
  }
  // ---- followers: wait (bounded) for the leader's inverse table
  bool have_inverse = false;
  if (split && !leader) {
    if (tid == 0) {
      int ok = 0;
      for (int it = 0; it < CC_POLLS; ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) {
          ok = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 forgets what it may hold of the table
      leader_ok = ok;
    }
    __syncthreads();
    have_inverse = leader_ok != 0;  // workgroup-uniform
    if (have_inverse) {
      // XT[c][nn] = Linv[nn][c] = LinvT[c][nn]: the published Linv^T block, row by row
      for (int idx = tid; idx < 64 * 16; idx += 256) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        *reinterpret_cast<f32x4*>(&XT[row * S + c4]) = *reinterpret_cast<const f32x4*>(LiT + row * 64 + c4);
      }
    }
  }
  // ---- leader / redundant form / time-out: T = A[S, S] - sum_t L[S, t] L[S, t]^T -> LDS, factorise, invert
  if (!have_inverse) {
    const float* pa = Ai + (S0 + 32 * ti + l31) * ld + 32 * kk;
    const float* pb = Ai + (S0 + 32 * tj + l31) * ld + 32 * kk;
    const f32x16 acc = cc_tile_product(pa, pb, k);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kk, col = 32 * tj + l31;
      L[row * S + col] = Ai[(S0 + row) * ld + S0 + col] - acc[r];
    }
    __syncthreads();
    if (wave == 0) chol_diag_factor_wave(L, LT, &progress, lane);
    else if (wave == 1) chol_diag_inverse_wave(L, XT, &progress, lane);
  }
  __syncthreads();

  // ---- the factor block and the inverse tables: the leader, or row block 0 of the redundant form.  The factor block does
  // NOT go into A[S, S]: other workgroups of this launch may still read the original block there (time-out path, column 0).
  // It goes, transposed, into the diagonal block of LT; chol_col_restore_kernel copies the diagonal blocks back into A.
  if (leader || (!split && rb == 0)) {
    chol_diag_writeback(L, XT, nullptr, ld, Li, LiT, tid, 256);
    float* ltd = LTm + (long)img * strideLT + S0 * (long)n + S0;
    for (int idx = tid; idx < 64 * 16; idx += 256) {
      const int row = idx >> 4, c4 = (idx & 15) * 4;
      f32x4 v;
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (c4 + u >= row) ? L[(c4 + u) * S + row] : 0.f;  // LT[row][c] = L[c][row], c >= row
      *reinterpret_cast<f32x4*>(ltd + (long)row * n + c4) = v;
    }
    // publish: every store of this workgroup retired, then ONE lane releases at agent scope and raises the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (hipcc may drop the wait behind buffer_wbl2: restate it)
      __hip_atomic_store(flag, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (leader) return;
  }

  // ---- L[R, S] = P Linv^T : tile (ti, tj) per wave.  Linv[nn][c] = X[nn][c] = XT[c][nn]  (XT[c][nn] = 0 for nn < c)
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(&Pb[(32 * ti + l31) * S + 32 * kk + 4 * q]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = 32 * kk + 4 * q + u, nn = 32 * tj + l31;
        const float xv = nn >= c ? XT[c * S + nn] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[u], xv, acc, 0, 0, 0);
      }
    }
    float* dst = Ai + (R0 + 32 * ti + 4 * kk) * ld + S0 + 32 * tj + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * ld] = acc[r];
    if (R0 < n) {  // rows of the matrix proper (not the right-hand sides): L^T for the backward substitution
      float* lt = LTm + (long)img * strideLT + (S0 + 32 * tj + l31) * (long)n + R0 + 32 * ti + 4 * kk;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(lt + 8 * g) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    }
  }
}

// A[S_k, S_k] <- lower triangle of (LT[S_k, S_k])^T for every block column k: the factor's diagonal blocks, upper part zero
__global__ __launch_bounds__(256) void chol_col_restore_kernel(float* __restrict__ A, long ld, long strideA,
                                                               const float* __restrict__ LTm, long strideLT, int n) {
  __shared__ float tile[64 * 65];
  const int k = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  const float* src = LTm + (long)img * strideLT + (64l * k) * n + 64l * k;
  float* dst = A + (long)img * strideA + (64l * k) * ld + 64l * k;
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    tile[r * 65 + c] = c >= r ? src[(long)r * n + c] : 0.f;  // (the strictly lower part of LT's block carries the hand-off flag)
  }
  __syncthreads();
  for (int idx = tid; idx < 64 * 64; idx += 256) dst[(long)(idx >> 6) * ld + (idx & 63)] = tile[(idx & 63) * 65 + (idx >> 6)];
}

int g_gp_col_leader = -1;  // roma_tuning("gp_col_leader", v): 1 = leader + followers (default), 0 = every workgroup factorises its own copy

// Block column k of the augmented system: A [batch][(n + d) x n] (ld = n), LT [batch][n x n].  `epoch`: any value that differs
// between consecutive solves on the same buffers (cholesky_solve_t counts its calls).
int chol_col_launch(float* A, long ld, long strideA, float* LT, long strideLT, int n, int d, float* Linv, float* LinvT, int k,
                    int nblk, int batch, unsigned epoch, hipStream_t s) {
  ROMA_REQUIRE(n % 64 == 0 && d % 64 == 0 && d >= 64 && ld % 4 == 0 && LT, "chol_col: n, d multiples of 64, d >= 64, ld of 4, LT");
  const int nrb = (n + d) / 64 - (k + 1);
  ROMA_REQUIRE(nrb >= 1 && k < nblk, "chol_col: no row block below the column");
  static const int leader_env = getenv("ROMA_GP_COL_LEADER") ? atoi(getenv("ROMA_GP_COL_LEADER")) : 1;
  const int use_leader = g_gp_col_leader >= 0 ? g_gp_col_leader : leader_env;
  const int gx = nrb + ((use_leader && k > 0) ? 1 : 0);
  hipLaunchKernelGGL(chol_col_kernel, dim3((unsigned)gx, (unsigned)batch), dim3(256), 0, s, A, ld, strideA, LT, strideLT, n, Linv,
                     LinvT, k, nblk, epoch & 0xffffffu, use_leader);
  ROMA_LAUNCH_CHECK();
  return 0;
}

int chol_col_restore_launch(float* A, long ld, long strideA, const float* LT, long strideLT, int n, int batch, hipStream_t s) {
  hipLaunchKernelGGL(chol_col_restore_kernel, dim3((unsigned)(n / 64), (unsigned)batch), dim3(256), 0, s, A, ld, strideA, LT,
                     strideLT, n);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
