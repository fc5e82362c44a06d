// DRAFT, NOT YET RUN ON HARDWARE (written at the end of round 1 with no GPU minutes left): stand-alone microbenchmark of
// the "8-phase" 256x256x64 bf16 NT GEMM main loop that DESIGN.md names as the next step for gemm.hip.  It exists so
// that the next round starts from a compiled candidate plus a race screen instead of from a blank page; nothing in the
// product links it.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm8p_draft gemm8p_draft.hip
//   ./gemm8p_draft [M N K [runs]]        (M, N multiples of 256, K multiple of 64, K >= 192)
//
// C[M,N] (bf16) = A[M,K] * W[N,K]^T, f32 accumulate.  Prints TFLOP/s, a sampled check against a scalar reference kernel
// and a run-to-run bitwise race screen.
//
// Schedule (cdna_hip_programming.md "256^2 8-phase template", restated for the 32x32x16 MFMA and the row-swizzled
// 128-byte LDS rows gemm.hip already uses):
//   * 8 waves = 2 wave groups (wr = wave / 4) x 4 column slices (wc = wave % 4); waves w and w + 4 share a SIMD and belong
//     to different groups.  Group 1 executes ONE extra s_barrier up front, so between any two consecutive barriers one
//     group is in its load block (ds_read fragments + one half-tile of LDS-DMA) and the other in its MFMA block.
//   * wave (wr, wc) owns C rows {64 wr + [0,64)} u {128 + 64 wr + [0,64)} and columns {32 wc + [0,32)} u {128 + 32 wc + [0,32)}
//     so that its four 64 x 32 quadrants read whole half-tiles: A0 = tile rows 0..127, A1 = 128..255, B0 / B1 likewise.
//   * one K tile = 4 phases:  P1 reads A0 + B0 -> quadrant (0,0);  P2 reads B1 -> (0,1);  P3 reads A1 -> (1,1);
//     P4 reads nothing -> (1,0).  Every phase: [reads][one half-tile DMA][vmcnt at P4] barrier, lgkmcnt(0), 8 MFMA, barrier.
//   * LDS-DMA, one half-tile (2 x 1 KiB per wave) per phase, two LDS buffers:
//       P1(k): B1(k+1)   P2(k): A1(k+1)   P3(k): A0(k+2)   P4(k): B0(k+2), then vmcnt(4)
//     - a region is re-staged >= 2 phases after its last read (A0, B0 last read in P1; B1 in P2; A1 in P3);
//     - the vmcnt(4) at P4(k) leaves only A0(k+2), B0(k+2) in flight, i.e. all of tile k+1 has landed; it sits before
//       P4's first barrier and the first read of tile k+1 is in the next phase (wait -> barrier -> read).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef unsigned short bf16_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int BM = 256, BN = 256, BK = 64, ROWB = 128;  // 64 bf16 per LDS row
constexpr int TILE_A = BM * ROWB, TILE_W = BN * ROWB, BUF = TILE_A + TILE_W;  // 64 KiB per K tile, two buffers

__device__ __forceinline__ unsigned pack2(float a, float b) {
  pk_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pk_bf16x2));
}

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#define DS_READ(REG, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(REG) : "v"(ADDR), "n"(OFF))

__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                        bf16_t* __restrict__ C, int M, int N, int K, int skip_store) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, h = lane >> 5;
  const int mt_n = N / BN;
  const long m0 = (long)(blockIdx.x / mt_n) * BM;
  const int n0 = (blockIdx.x % mt_n) * BN;
  const int nk = K / BK;

  // ---- DMA descriptors: a half-tile is 128 rows = 16 pieces of 8 rows; wave w stages pieces 2w, 2w + 1 of every half-tile
  const int r8 = lane >> 3, slot = lane & 7;
  const char* a_src[2][2];  // [half][piece]
  const char* w_src[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = hf * 128 + 8 * (2 * wave + j) + r8;
      const int chunk = slot ^ ((row >> 1) & 7);
      a_src[hf][j] = reinterpret_cast<const char*>(A + (m0 + row) * (long)K + chunk * 8);
      w_src[hf][j] = reinterpret_cast<const char*>(W + (long)(n0 + row) * K + chunk * 8);
    }
#define ISSUE_A(HF, KT)                                                                                          \
  {                                                                                                              \
    char* dst_ = smem + ((KT) & 1) * BUF + (HF) * 128 * ROWB + (2 * wave) * 1024;                                \
    glds16(a_src[HF][0] + (long)(KT) * (BK * 2), dst_);                                                          \
    glds16(a_src[HF][1] + (long)(KT) * (BK * 2), dst_ + 1024);                                                   \
  }
#define ISSUE_W(HF, KT)                                                                                          \
  {                                                                                                              \
    char* dst_ = smem + ((KT) & 1) * BUF + TILE_A + (HF) * 128 * ROWB + (2 * wave) * 1024;                       \
    glds16(w_src[HF][0] + (long)(KT) * (BK * 2), dst_);                                                          \
    glds16(w_src[HF][1] + (long)(KT) * (BK * 2), dst_ + 1024);                                                   \
  }

  // ---- fragment addresses: row = block_row0 + l31, 16-byte slot (2g + h) ^ ((row >> 1) & 7); block_row0 % 32 == 0
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const int sw = (l31 >> 1) & 7;
  unsigned rd[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) rd[g] = (unsigned)(l31 * ROWB + (((2 * g + h) ^ sw) << 4));
  const unsigned a_row0 = (unsigned)((64 * wr) * ROWB);            // + mh * 128 rows + mt * 32 rows
  const unsigned w_row0 = (unsigned)(TILE_A + (32 * wc) * ROWB);   // + nh * 128 rows

  f32x16 acc[4][2];  // [mh * 2 + mt][nh]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  u32x4 af[2][4], bf0[4], bf1[4];  // A fragments [mt][k-group] of the current 64-row half, W fragments of both column halves

#define READ_A(MH, SB)                                                                        \
  _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int g = 0; g < 4; ++g) \
      DS_READ(af[mt][g], (SB) + a_row0 + rd[g], ((MH) * 128 + mt * 32) * ROWB);
#define READ_B(BF, NH, SB) \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) DS_READ(BF[g], (SB) + w_row0 + rd[g], ((NH) * 128) * ROWB);
  // D[n][m] += W[n][k] A[m][k]: W fragment as the first operand, so a lane owns 4 consecutive n of one m (gemm.hip)
#define MFMA_Q(MH, NH, BF)                                                                                       \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                   \
      acc[(MH) * 2 + mt][NH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, BF[g]),      \
                                                                       __builtin_bit_cast(bf16x8_t, af[mt][g]),  \
                                                                       acc[(MH) * 2 + mt][NH], 0, 0, 0);
#define PHASE_SYNC_MFMA(MF)                      \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  WAIT_LGKM(0);                                  \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_setprio(1);                 \
  MF;                                            \
  __builtin_amdgcn_s_setprio(0);                 \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: tile 0 complete, A0 / B0 of tile 1 under way
  ISSUE_A(0, 0) ISSUE_W(0, 0) ISSUE_W(1, 0) ISSUE_A(1, 0)
  ISSUE_A(0, 1) ISSUE_W(0, 1)
  WAIT_VM(4);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier behind group 0 from here on

  for (int kt = 0; kt < nk; ++kt) {
    const unsigned sb = lds0 + (unsigned)((kt & 1) * BUF);
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
    // P1: A0 + B0 -> (0,0); stage B1(k+1)
    READ_B(bf0, 0, sb)
    __builtin_amdgcn_sched_barrier(0);
    READ_A(0, sb)
    if (n1) ISSUE_W(1, kt + 1)
    PHASE_SYNC_MFMA(MFMA_Q(0, 0, bf0))
    // P2: B1 -> (0,1); stage A1(k+1)
    READ_B(bf1, 1, sb)
    if (n1) ISSUE_A(1, kt + 1)
    PHASE_SYNC_MFMA(MFMA_Q(0, 1, bf1))
    // P3: A1 -> (1,1); stage A0(k+2)
    READ_A(1, sb)
    if (n2) ISSUE_A(0, kt + 2)
    PHASE_SYNC_MFMA(MFMA_Q(1, 1, bf1))
    // P4: no reads -> (1,0); stage B0(k+2); everything of tile k+1 must have landed before this phase's first barrier
    if (n2) {
      ISSUE_W(0, kt + 2)
      WAIT_VM(4);
    } else {
      WAIT_VM(0);
    }
    PHASE_SYNC_MFMA(MFMA_Q(1, 0, bf0))
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // group 0 catches up with group 1's extra barrier

  // ---- plain epilogue (the product kernel's staged row writer is a separate piece of work)
  if (skip_store) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int mh = q >> 1, mt = q & 1;
    const long m = m0 + mh * 128 + 64 * wr + mt * 32 + l31;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + nh * 128 + 32 * wc + 8 * rg + 4 * h;
        uint2 pk;
        pk.x = pack2(acc[q][nh][4 * rg + 0], acc[q][nh][4 * rg + 1]);
        pk.y = pack2(acc[q][nh][4 * rg + 2], acc[q][nh][4 * rg + 3]);
        *reinterpret_cast<uint2*>(C + m * (long)N + n) = pk;
      }
  }
}

__global__ void ref_kernel(const bf16_t* A, const bf16_t* W, float* out, const int* mi, const int* ni, int cnt, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const bf16_t* a = A + (long)mi[i] * K;
  const bf16_t* w = W + (long)ni[i] * K;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += __uint_as_float((unsigned)a[k] << 16) * __uint_as_float((unsigned)w[k] << 16);
  out[i] = s;
}

static bf16_t f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (bf16_t)(u >> 16);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  const int runs = argc > 4 ? atoi(argv[4]) : 20;
  if (M % BM || N % BN || K % BK || K < 3 * BK) {
    fprintf(stderr, "M, N multiples of 256, K multiple of 64 and >= 192\n");
    return 1;
  }
  std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
  uint64_t s = 0x9e3779b97f4a7c15ull;
  auto rnd = [&]() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (float)((s >> 40) & 0xffff) / 32768.f - 1.f;  // uniform [-1, 1): quote random-data numbers only
  };
  for (auto& v : hA) v = f2bf(rnd());
  for (auto& v : hW) v = f2bf(rnd());
  bf16_t *dA, *dW, *dC, *dC2;
  CHECK(hipMalloc(&dA, hA.size() * 2));
  CHECK(hipMalloc(&dW, hW.size() * 2));
  CHECK(hipMalloc(&dC, (size_t)M * N * 2));
  CHECK(hipMalloc(&dC2, (size_t)M * N * 2));
  CHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  const int lds = 2 * BUF;
  CHECK(hipFuncSetAttribute((const void*)gemm8p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const dim3 grid((M / BM) * (N / BN)), block(512);
  hipLaunchKernelGGL(gemm8p_kernel, grid, block, lds, 0, dA, dW, dC, M, N, K, 0);
  CHECK(hipDeviceSynchronize());

  // sampled reference
  const int cnt = 4096;
  std::vector<int> mi(cnt), ni(cnt);
  for (int i = 0; i < cnt; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    mi[i] = (int)(s % (uint64_t)M);
    ni[i] = (int)((s >> 32) % (uint64_t)N);
  }
  int *dmi, *dni;
  float* dref;
  CHECK(hipMalloc(&dmi, cnt * 4)); CHECK(hipMalloc(&dni, cnt * 4)); CHECK(hipMalloc(&dref, cnt * 4));
  CHECK(hipMemcpy(dmi, mi.data(), cnt * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dni, ni.data(), cnt * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ref_kernel, dim3((cnt + 255) / 256), dim3(256), 0, 0, dA, dW, dref, dmi, dni, cnt, K);
  std::vector<float> href(cnt);
  std::vector<bf16_t> hC((size_t)M * N);
  CHECK(hipMemcpy(href.data(), dref, cnt * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
  double worst = 0;
  int bad = 0;
  for (int i = 0; i < cnt; ++i) {
    unsigned u = (unsigned)hC[(size_t)mi[i] * N + ni[i]] << 16;
    float got;
    memcpy(&got, &u, 4);
    const double err = fabs((double)got - href[i]), tol = 0.01 * fabs(href[i]) + 0.05;
    if (err > tol) ++bad;
    if (err > worst) worst = err;
  }
  printf("check: %d / %d sampled outputs outside tolerance, max |err| %.4f\n", bad, cnt, worst);

  // race screen: bitwise run-to-run
  int diff_runs = 0;
  for (int r = 0; r < runs; ++r) {
    hipLaunchKernelGGL(gemm8p_kernel, grid, block, lds, 0, dA, dW, dC2, M, N, K, 0);
    CHECK(hipDeviceSynchronize());
    std::vector<bf16_t> h2((size_t)M * N);
    CHECK(hipMemcpy(h2.data(), dC2, h2.size() * 2, hipMemcpyDeviceToHost));
    if (memcmp(h2.data(), hC.data(), h2.size() * 2) != 0) ++diff_runs;
  }
  printf("race screen: %d / %d runs differ bitwise from the first\n", diff_runs, runs);

  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int skip = 0; skip < 2; ++skip) {
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < runs; ++r) hipLaunchKernelGGL(gemm8p_kernel, grid, block, lds, 0, dA, dW, dC2, M, N, K, skip);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= runs;
    printf("M=%d N=%d K=%d %s: %.3f ms  %.0f TFLOP/s\n", M, N, K, skip ? "no stores" : "with stores", ms,
           2.0 * M * N * K / (ms * 1e-3) / 1e12);
  }
  return bad != 0 || diff_runs != 0;
}
