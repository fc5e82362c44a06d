// Flash-style multi-head attention on MFMA (online softmax, no N x N score matrix in HBM).
// Layouts (written by the QKV GEMM epilogue, gemm.h EPI_QKV):
//   q, k : [B, heads, Npad, hd]   (q pre-scaled by 1/sqrt(hd))
//   vt   : [B, heads, hd, Npad]   (V transposed, so both MFMA operands are K-contiguous)
//   out  : [B*N, heads*hd] row-major (ldo)  - token-major, ready for the proj GEMM
// Npad is a multiple of 128; rows >= N are zero and are masked as keys.
#pragma once
#include "common.h"

namespace roma {
struct AttnArgs {
  const void *q = nullptr, *k = nullptr, *vt = nullptr;
  void* out = nullptr;
  int B = 0, heads = 0, N = 0, npad = 0, hd = 0;
  long ldo = 0;
  int in_dt = 0, out_dt = 0;  // DT_F32 / DT_BF16
  int exp2_domain = 0;        // bf16 kernel: q was pre-scaled by log2(e)/sqrt(hd), softmax uses v_exp_f32 (2^x) directly
  int xcd_map = 1;            // set by attention_launch (g_attn_xcd_map): workgroup -> work item order, see attention.hip
};
extern int g_attn_xcd_map;
extern int g_attn_exp2;
int attention_launch(const AttnArgs& a, hipStream_t stream);
}  // namespace roma
