// Weight-stationary 1x1 convolution for the stride-4 ConvRefiner (C = 576 = the padded 569 channels of
// romatch/models/matcher.py:92-122,175-176): C[M, 576] = act(A[M, 576] . W[576, 576]^T + bias), 16-bit in / out, f32 accumulate.
// 18 of the 36 refiner GEMM launches of a step and their most expensive half: K = N = 576 puts the arithmetic intensity at
// the ridge (288 FLOP / B) and leaves a 256 x 192 GEMM tile with 9 K tiles per output tile - 40 % of gemm6p's time on these
// shapes is prologue / epilogue, its loop feeds 110 FLOP per staged byte, and it runs at 0.24 - 0.29 of the MFMA peak, like
// the vendor library (profiles/r03_v8_vendor_gemm_yardstick.log).
//
// K = 576 is exactly the K of conv3x3_c64 (9 taps x 64 channels), so the anatomy of conv64.hip applies: a wave keeps
// W[32 couts][576 k] in 144 VGPRs for its whole life (36 MFMA A operands), only PIXEL rows move - once per workgroup through
// LDS by LDS-DMA, 256 FLOP per staged byte - and a chunk of pixels is one uninterrupted chain of 36 MFMAs per wave with
// nothing to prologue or drain.
//
//   * N = 576 = 18 blocks of 32 couts = 2.25 x the 8 blocks an 8-wave workgroup can hold.  A workgroup therefore has SIX
//     waves (6 blocks, 192 couts) and three workgroups serve one pixel stream; the three sit on the same XCD (workgroup b runs
//     on XCD b % 8, the group is three consecutive b / 8) so that the stream crosses HBM -> L2 once.  Two of a CU's SIMDs host
//     two waves, two host one: 75 % of the matrix pipe at best - the price of 18 = 3 x 6.  (Round 4 measured the alternative
//     first: 8-wave workgroups in groups of nine serving four streams, three of the nine straddling two streams.  Every wave
//     held useful couts, but the straddling workgroups stage twice the bytes per MFMA and the launch ran at their pace:
//     637 / 818 TFLOP/s at M = 746 496 / 313 600, profiles/r04_v13_ws1x1_first.log.)
//   * a step = one chunk of 32 pixels x 1152 B = 36 KiB, CONTIGUOUS in memory (lda = 576): 36 DMA instructions of one KiB, six
//     per wave, and 36 MFMAs per wave.  The chunks go through a ring of FOUR slots (144 KiB): during step t the waves issue
//     the DMA of chunk t + 3 BETWEEN their MFMAs (one piece per six k-steps), into the slot chunk t - 1 left - so a chunk has
//     two full steps (~2 us) to arrive.  (Two measured dead ends, profiles/r04_v13b_ws1x1_second.log: all pieces of a
//     72 KiB double-chunk step issued as a burst behind the barrier cost every wave ~3 000 cycles of texture-path queueing
//     before its first MFMA; interleaved but only ONE step ahead, the last pieces were awaited ~300 cycles after their issue.)
//   * LDS position (pixel p, 16-byte chunk c) holds source chunk c ^ ((p >> 1) & 7): pixel rows are 1152 B = 4.5 bank rows
//     apart, so the 16 pixels of a ds_read_b128 lane group alternate between two bank-row halves (p & 1) and the XOR spreads
//     each half over its 8 slots - conflict free, the swizzle of conv64.hip (permutation on the DMA source address).
//   * per step: counted wait for the own pieces of chunk t (allowance = the younger LOADS only: the pieces of chunks t + 1,
//     t + 2, exact also at the stream's end), barrier, store the outputs of step t - 1 (kept packed in 8 registers), multiply
//     with the DMA of chunk t + 3 interleaved.  Fragment reads run 3 k-steps ahead of their MFMAs (counted lgkmcnt).
//   * output: D[cout][pixel] + bias (+ ReLU), v_permlane32_swap pairs the two half-waves so that a lane holds 8 consecutive
//     couts (16 bytes) of one pixel (the epilogue of conv64.hip); the six waves then exchange the 32 x 192 tile through LDS
//     and store 384-byte runs.
//   * same k order as the GEMM kernels (one accumulator per output, k ascending): bit-identical to gemm6p / gemm_kernel,
//     which is the race screen (tests/test_gpu_ops.py).
#include "gemm.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gemm_device.h"

namespace roma {

int g_ws1x1_mode = -1;  // roma_tuning("ws1x1", v): 1 = this kernel for the C = 576 refiner GEMMs (default), 0 = gemm6p; -1 = env ROMA_WS1X1

#define WS_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

constexpr int WS_K = 576, WS_N = 576, WS_PX = 32;     // channels, pixels per slot
constexpr int WS_ROWB = WS_K * 2;                      // 1152 bytes per pixel row
constexpr int WS_SLOT = WS_PX * WS_ROWB;               // 36 KiB
constexpr int WS_LDS = 4 * WS_SLOT;                    // ring of four chunks: 144 KiB (+ bias)

template <int ACT>
__global__ __launch_bounds__(384, 2) void ws1x1_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                       const float* __restrict__ bias, bf16_t* __restrict__ C, long nchunks,
                                                       long chunks_per_stream, int streams_per_xcd, int dbg) {
  // dbg (roma_tuning "ws1x1" bits, measurements only): 2 = no output stores, 4 = no DMA after the first three chunks, 8 = no MFMA
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];  // [4 slots][32 px][1152 B], then bias f32[576]
#ifdef ROMA_TOOLS_BUILD
  const int dbg_ = dbg;
#else
  constexpr int dbg_ = 0;  // the ablation bits exist in tools builds only (make TOOLS=1)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)lds);

  // ---- who am I: XCD x = b % 8, local index i = b / 8 -> stream i / 3 of this XCD, third i % 3 of the couts
  const int xcd = blockIdx.x % 8, li = blockIdx.x / 8;
  const int sl = li / 3, part = li - 3 * sl;
  if (sl >= streams_per_xcd) return;
  const long stream = (long)xcd * streams_per_xcd + sl;
  const int nb = 6 * part + wave;  // this wave's cout block
  // stream S covers chunks [S cps, min((S + 1) cps, nchunks)), one per step
  const long st0 = stream * chunks_per_stream;
  const long end0 = std::min(st0 + chunks_per_stream, nchunks);
  const long nsteps = std::max(end0 - st0, 0l);

  // ---- this wave's weights: W[32 nb + l31][16 ks + 8 h .. + 8), ks = 0 .. 35 (A operands: rows = couts)
  u32x4 wreg[36];
  {
    const bf16_t* wp = W + (long)(32 * nb + l31) * WS_K + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 36; ++ks) wreg[ks] = *reinterpret_cast<const u32x4*>(wp + 16 * ks);
  }
  float* bias_s = reinterpret_cast<float*>(lds + WS_LDS);
  for (int i = tid; i < WS_N; i += 384) bias_s[i] = bias[i];

  // ---- fragment read addresses: pixel l31, chunk (2 ks + h) ^ sw, sw = (l31 >> 1) & 7: the low three chunk bits depend on
  // ks & 3, the rest (ks >> 2) * 128 bytes is an immediate
  unsigned rd[4];
  {
    const int sw = (l31 >> 1) & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) rd[i] = lds0 + (unsigned)(l31 * WS_ROWB + ((((2 * i + h) & 7) ^ sw) << 4));
  }
  // ---- DMA: piece r = 6 wave + i (KiB r of the chunk); lane -> LDS 16-byte position S = 64 r + lane = 72 p + c', source
  // chunk c' ^ ((p >> 1) & 7) of pixel p: one contiguous KiB per instruction
  unsigned soff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int r = 6 * wave + i;
    const int S = 64 * r + lane, p = S / 72, cp = S - 72 * p;
    soff[i] = (unsigned)(p * WS_ROWB + ((cp & ~7) | ((cp & 7) ^ ((p >> 1) & 7))) * 16);
  }
  // weights, bias: nothing in flight before the DMA stream starts - as a builtin, i.e. in a form hipcc's wait-count pass sees,
  // and with every weight register consumed right behind it: otherwise the pass carries "global load pending" on the weight
  // registers into the loop and puts an s_waitcnt vmcnt(0) in front of the first MFMA of EVERY step (the whole ring drained)
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
#pragma unroll
  for (int ks = 0; ks < 36; ++ks) asm volatile("" : "+v"(wreg[ks]));

  // piece I (0 .. 5) of this wave for chunk CH (a chunk index of the stream, wave-uniform) into ring slot CH & 3
  // (round 6: through a buffer descriptor - SGPR base, the lane's 32-bit offset inside the chunk, the chunk offset in an SGPR -
  //  instead of a 64-bit lane address rebuilt per piece; gemm8p.hip)
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)(nchunks * (long)WS_SLOT), 0x00020000);
#define WS_ISSUE_PIECE(I, CH)                                                                                         \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(lds + (int)((CH) & 3) * WS_SLOT + (6 * wave + (I)) * 1024), \
                                           16, (int)soff[I], (int)((CH) * (long)WS_SLOT), 0, 0);

  u32x4 fa[4];  // reads run three k-steps ahead of their MFMAs
#define WS_READ(SB, KS) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[(KS) & 3]) : "v"(rd[(KS) & 3] + (SB)), "n"(((KS) >> 2) * 128))
  // the read of k-step KS is the oldest in flight; N = younger reads allowed to stay in flight
#define WS_WAIT1(KS, N) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fa[(KS) & 3]) : "n"(N) : "memory")

  u32x4 outp[2];  // packed outputs of the previous step (stored at the top of the next one)
  long out_chunk = -1;
  // Output path.  A wave's MFMA block is 32 couts = 64 bytes per pixel; stored as it is (two 16-byte pieces per lane pair)
  // every 128-byte line of C is written in four separate pieces by two waves at different times - 115 of the 757 us of the
  // M = 746 496 launch were the stores (ablation, profiles/r04_v14_ws1x1_ablation.log).  So the six waves exchange the step's
  // outputs through an LDS tile [32 pixels][192 couts] (400-byte pitch: conflict-free 16-byte writes) and every thread stores
  // two 16-byte chunks of 384-byte runs.  Barrier A (ring) - write tile - barrier B - read tile: the next write sits behind
  // the next barrier A, which no wave passes before it has read.
  // The stores are inline asm: hipcc orders a VMEM store it can see against the NEXT write of its data registers with
  // s_waitcnt vmcnt(0) (loads and stores are "unordered" to its wait-count pass), which would drain the DMA ring every step.
  // The hardware only needs the store's operands read: two wait states behind a store of more than 64 bits (the s_nop).
  constexpr int WS_TPITCH = 400;
  const unsigned tile0 = lds0 + WS_LDS + WS_N * 4;
  const unsigned tw = tile0 + (unsigned)(l31 * WS_TPITCH + 64 * wave + 16 * h);          // + 32 P
  int cid0 = tid, cid1 = tid + 384;                                                       // 16-byte chunks of the tile
  const unsigned tr0 = tile0 + (unsigned)((cid0 / 24) * WS_TPITCH + (cid0 % 24) * 16);
  const unsigned tr1 = tile0 + (unsigned)((cid1 / 24) * WS_TPITCH + (cid1 % 24) * 16);
  const long go0 = (long)(cid0 / 24) * (WS_N * 2) + part * 384 + (cid0 % 24) * 16;         // byte offsets inside a chunk of C
  const long go1 = (long)(cid1 / 24) * (WS_N * 2) + part * 384 + (cid1 % 24) * 16;
#define WS_STORE_OUT()                                                                                                \
  {                                                                                                                   \
    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:32" ::"v"(tw), "v"(outp[0]), "v"(outp[1]) : "memory"); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    __builtin_amdgcn_s_barrier();                                                                                     \
    u32x4 q0_, q1_;                                                                                                   \
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"                               \
                 : "=&v"(q0_), "=&v"(q1_)                                                                             \
                 : "v"(tr0), "v"(tr1)                                                                                 \
                 : "memory");                                                                                         \
    if (!(dbg_ & 2)) {                                                                                                \
      char* cb_ = reinterpret_cast<char*>(C) + out_chunk * (long)(WS_PX * WS_N * 2);                                  \
      asm volatile("global_store_dwordx4 %0, %2, off\n\tglobal_store_dwordx4 %1, %3, off\n\ts_nop 1"                 \
                   :                                                                                                  \
                   : "v"(cb_ + go0), "v"(cb_ + go1), "v"(q0_), "v"(q1_)                                               \
                   : "memory");                                                                                       \
    }                                                                                                                 \
  }

  // chunks st0 .. st0 + 2 (ring slots (st0 + k) & 3) before the first step
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (st0 + k < end0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) WS_ISSUE_PIECE(i, st0 + k)
    }

  // Round 6: steps with the ring full run the STEADY copy of the body, the last three the general one (ws1x1_step.inc)
  long c = st0;
  for (; c + 3 < end0 && !(dbg_ & 4); ++c) {
#define WS_STEADY 1
#include "ws1x1_step.inc"
#undef WS_STEADY
  }
  for (; c < end0; ++c) {
#define WS_STEADY 0
#include "ws1x1_step.inc"
#undef WS_STEADY
  }
  // the outputs of the last step
  if (out_chunk >= 0) {
    __builtin_amdgcn_s_barrier();  // the barrier A of a step that does not exist: every wave has read the tile of the step before
    WS_STORE_OUT()
  }
  WS_WAIT_VM0();
#undef WS_STORE_OUT
#undef WS_WAIT1
#undef WS_READ
#undef WS_ISSUE_PIECE
}

static int ws_mode() {
  static const int env = getenv("ROMA_WS1X1") ? atoi(getenv("ROMA_WS1X1")) : 1;
  return g_ws1x1_mode >= 0 ? g_ws1x1_mode : env;
}

template <int ACT>
static int launch_ws(const GemmArgs& a, hipStream_t stream) {
  const long nchunks = a.M / WS_PX;
  const int streams_per_xcd = 10;  // 30 of an XCD's 32 CUs: ten groups of three workgroups
  const long nstreams = 8l * streams_per_xcd;
  long cps = (nchunks + nstreams - 1) / nstreams;
  const size_t lds = (size_t)WS_LDS + WS_N * sizeof(float) + 32 * 400;  // ring + bias + output tile = 162 560 B
  char pname[96];
  snprintf(pname, sizeof pname, "ws1x1_kernel<" ROMA_H16_NAME ",%s>", ACT == ACT_RELU ? "relu" : "none");
  ProfScope ps(pname, 2.0 * (double)a.M * (a.n_alg > 0 ? a.n_alg : a.N) * (a.k_alg > 0 ? a.k_alg : a.K), "flop", stream);
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ws1x1_kernel<ACT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((ws1x1_kernel<ACT>), dim3((unsigned)(8 * 3 * streams_per_xcd)), dim3(384), lds, stream, (const bf16_t*)a.A,
                     (const bf16_t*)a.W, a.bias, (bf16_t*)a.C, nchunks, cps, streams_per_xcd, ws_mode() & ~1);
  ROMA_LAUNCH_CHECK();
  return 0;
}

// 0 = launched, 1 = not this kernel's problem, < 0 = error.  Called by gemm8p_try_launch ahead of gemm6p.
int ws1x1_try_launch(const GemmArgs& a, hipStream_t stream) {
  if (!(ws_mode() & 1)) return 1;
  if (a.N != WS_N || a.K != WS_K || a.lda != WS_K || a.ldw != WS_K || a.ldc != WS_N) return 1;
  if (a.in_dt != DT_BF16 || a.out_dt != DT_BF16 || a.batch != 1 || a.batch2 != 1 || a.conv_c > 0 || a.mode != EPI_STD) return 1;
  if (a.res || a.res_bf16 || a.scale || a.qkv_pad || a.alpha != 1.0f || !a.bias) return 1;
  if (a.act != ACT_NONE && a.act != ACT_RELU) return 1;
  if (a.M % WS_PX != 0 || a.M < 64 * 1024) return 1;  // whole 32-pixel chunks; small problems stay on the tile kernels
  if (((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.C) | reinterpret_cast<uintptr_t>(a.W)) & 15) != 0) return 1;
  if (a.dbg & 0x3ff) return 1;  // tuning experiments address the tile kernels
#ifndef ROMA_TOOLS_BUILD
  ROMA_REQUIRE(!(ws_mode() & ~1), "ws1x1: the ablation bits (2 / 4 / 8 of roma_tuning ws1x1) exist in tools builds only (make TOOLS=1)");
#endif
  if (a.act == ACT_RELU) return launch_ws<ACT_RELU>(a, stream);
  return launch_ws<ACT_NONE>(a, stream);
}

}  // namespace roma
