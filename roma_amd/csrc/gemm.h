// MFMA GEMM family:  C[M,N] = epilogue( alpha * A[M,K] . W[N,K]^T )   (both operands K-contiguous)
//   * f32 inputs  -> v_mfma_f32_32x32x2_f32   (exact f32, parity mode)
//   * bf16 inputs -> v_mfma_f32_32x32x16_bf16 (throughput mode), f32 accumulate
//   * A may be an implicit im2col view of an NHWC tensor (3x3, pad 1)  -> VGG convolutions
//   * epilogues: bias / activation / per-column scale / residual, QKV head scatter (+V transposed),
//     cosine-kernel Gram matrix (GP), alpha=-1 accumulate (Cholesky trailing update / TRSM).
#pragma once
#include "common.h"

namespace roma {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };
enum { EPI_STD = 0, EPI_QKV = 1, EPI_COSK = 2 };
enum { DT_F32 = 0, DT_BF16 = 1 };

struct GemmArgs {
  const void* A = nullptr;  // [M,K] (lda) or NHWC tensor when conv_c > 0
  const void* W = nullptr;  // [N,K] (ldw)
  void* C = nullptr;        // [M,N] (ldc)
  long lda = 0, ldw = 0, ldc = 0;
  long sA = 0, sW = 0, sC = 0;  // batch strides in elements
  int M = 0, N = 0, K = 0, batch = 1;
  // second batch level (gemm_kernel only; batch2 > 1 keeps the problem off the persistent / 8-phase paths): item z of the
  // batch * batch2 launched uses offsets (z / batch2) * s? + (z % batch2) * s?2 - e.g. (image, diagonal block) of the GP solve
  int batch2 = 1;
  long sA2 = 0, sW2 = 0, sC2 = 0, sR2 = 0;
  int in_dt = DT_F32, out_dt = DT_F32;
  float alpha = 1.f;
  const float* bias = nullptr;   // [N]
  const float* scale = nullptr;  // [N]
  const float* res = nullptr;    // f32 [M,N] (ldr), may alias C when C is f32
  long ldr = 0, sR = 0;
  // bf16 residual stream (DINOv2 in throughput mode): bf16 [M,N] with the same ldr / sR, may alias C.  Requires bf16
  // output, EPI_STD and 16-byte aligned rows (the staged row-writer adds it while it streams the tile out).
  const void* res_bf16 = nullptr;
  int act = ACT_NONE;
  int mode = EPI_STD;
  int lower_only = 0;  // skip tiles strictly above the diagonal (square problems)
  // implicit 3x3 convolution view of A (NHWC, pad 1); M = B*H*W, K = 9*conv_c
  int conv_h = 0, conv_w = 0, conv_c = 0;
  // K order of the convolution's weight rows: 0 = tap major, k = (ky*3 + kx) * Cin + ci (the operator entry point's layout);
  // 1 = slab major, k = ((ci / 64) * 9 + ky*3 + kx) * 64 + ci % 64 (Cin % 64 == 0): the nine taps of one 64-channel slab are
  // consecutive K tiles, so a workgroup re-reads the same (256 + 2 W) pixels x 128 bytes nine times in a row - a working set
  // the XCD's L2 holds - instead of sweeping all Cin channels of its pixels once per tap (round 5; the model packs the VGG
  // layers with Cout >= 256 this way in the 16-bit modes)
  int conv_korder = 0;
  // EPI_QKV
  void *q = nullptr, *k = nullptr, *vt = nullptr;
  int heads = 0, hd = 0, ntok = 0, npad = 0;
  float qscale = 1.f;
  int qkv_pad = 0;  // set by gemm_launch: GEMM rows are (image, token) with npad tokens per image (rows >= ntok read zeros)
  // EPI_COSK:  exp((acc/(nx[m]*ny[n]+1e-6) - 1) * inv_t) (+ diag_add on m==n)
  const float *nx = nullptr, *ny = nullptr;
  long sNx = 0, sNy = 0;
  float inv_t = 0.f, diag_add = 0.f;
  int m_alg = 0;  // set by gemm_launch with qkv_pad: the caller's (algorithmic) M, for the FLOP count of the profile scope
  // un-padded problem width / depth for the FLOP count of the profile scope (0 = N / K): the refiner 1x1 convolutions run
  // on channel counts padded to whole 128-byte rows (1137 -> 1152, 569 -> 576, 1377 -> 1408); the roofline figure counts
  // the reference's channels, not the zero padding
  int n_alg = 0, k_alg = 0;
  int walk_gm = 0;  // gemm8p: tile rows per group of the persistent walk (0 / 1 = row major; set by gemm8p_try_launch)
  int dbg = 0;  // tuning experiments only (ROMA_GEMM_DBG): 1 = skip output stores, 2 = skip the K loop
};

// Launches on `stream`; returns 0 or a negative error code (message via roma_last_error()).
int gemm_launch(const GemmArgs& a, hipStream_t stream);
// 8-phase 256 x 256 bf16 kernel (gemm8p.hip): 0 = launched, 1 = not its problem (gemm.hip runs it), < 0 = error
int gemm8p_try_launch(const GemmArgs& a, hipStream_t stream);
// its 256 x 192 sibling (gemm6p.hip), called by gemm8p_try_launch once the common preconditions hold
int gemm6p_try_launch(const GemmArgs& a, hipStream_t stream);
// weight-stationary 1x1 for N = K = 576 (ws1x1.hip), tried ahead of gemm6p; same return convention
int ws1x1_try_launch(const GemmArgs& a, hipStream_t stream);
extern int g_ws1x1_mode;  // roma_tuning("ws1x1", v): 1 on (default), 0 off, -1 = environment ROMA_WS1X1
extern int g_gemm8p_sched;    // gemm8p K-loop schedule: 1 = k-half phases, 0 = quadrant phases, -1 = environment ROMA_GEMM8P_SCHED (default 1)
extern int g_gemm8p_maxwg;
extern int g_gemm8p_walk;    // gemm8p walk group height (tools: tools/bench_gemm_walk.py), -1 = the dispatcher's choice
int gemm8p_trace_read(unsigned* host, long nbytes);  // phase trace of the last ablation-build launch (tools/bench_gemm_ablation.py)
extern int g_gemm_tuning[2];  // process-wide A/B switches (roma_tuning): [0] use gemm8p (-1 = env ROMA_GEMM8P, default on), [1] dbg bits

}  // namespace roma
