// HBM-bound kernels of the match() path (everything that is not a GEMM / attention / local-corr).
// All activations are channels-last; `dt` arguments are DT_F32 / DT_BF16 (gemm.h).
#pragma once
#include "common.h"

namespace roma {

// LayerNorm over the last dim (D multiple of 512, <= 2048): x f32 [M,D] -> out [M,D] (dt_out)
int layernorm_launch(const float* x, const float* w, const float* b, void* out, long M, int D, float eps,
                     int dt_out, hipStream_t s);
// same with a typed input: dt_in DT_F32, or DT_BF16 (bf16 residual stream; dt_out must then be DT_BF16)
int layernorm_launch_dt(const void* x, int dt_in, const float* w, const float* b, void* out, long M, int D, float eps,
                        int dt_out, hipStream_t s);

// First VGG layer: NCHW f32 image -> conv3x3(3->64, pad 1) + folded BN + ReLU -> NHWC (dt_out)
// w: [27][64] (tap-major: (ci*9+ky*3+kx), cout fastest), bias [64]
int conv3x3_c3_launch(const float* img, const float* w, const float* bias, void* out, int B, int H, int W,
                      int dt_out, hipStream_t s);

// im2col of the first layer: NCHW f32 image -> A[B*H*W, 32], k = ci*9 + ky*3 + kx (zero padding, k >= 27 zero)
int im2col3x3_c3_launch(const float* img, void* out, int B, int H, int W, int dt_out, hipStream_t s);

// MaxPool 2x2 stride 2 (floor), NHWC
int maxpool2x2_launch(const void* in, void* out, int B, int H, int W, int C, int dt, hipStream_t s);

// DINOv2 patchify: NCHW f32 image -> A[B*th*tw, Kpad] with k = c*196 + ky*14 + kx (zero padded to Kpad)
int im2col_patch14_launch(const float* img, void* out, int B, int H, int W, int Kpad, int dt_out, hipStream_t s);

// tokens[b,0,:] = cls + pos[0];  tokens[b,1+t,:] = patch[b,t,:] + pos[1+t]     (all f32, D columns)
int assemble_tokens_launch(const float* patch, const float* cls, const float* pos, float* tokens, int B, int T,
                           int D, hipStream_t s);

// rows x cols strided copy with dtype conversion: out[r*ldo + c] = in[r*ldi + c]
int copy2d_launch(const void* in, long ldi, int dt_in, void* out, long ldo, int dt_out, long rows, int cols,
                  hipStream_t s);

// bfloat16 bits -> this build's 16-bit format, n elements (n % 4 == 0); identity in the bf16 build
int convert_from_bf16_launch(const void* in, void* out, long n, hipStream_t s);

// L2 norm of each row: in [M, C] (ld, dt) -> norms f32 [M]
int rownorm_launch(const void* in, long ld, int dt, float* norms, long M, int C, hipStream_t s);

// GP Fourier basis, transposed: Ft[d, j] = cos(8*pi*(w[d,0]*x_j + w[d,1]*y_j + b[d])), j over the h x w grid
// (row-major, x fastest); columns j >= h*w are zero.  Ft: [Dg, npad]
int gp_basis_launch(const float* w, const float* b, float* Ft, int Dg, int h, int wdt, int npad, hipStream_t s, int copies = 1,
                    long stride = 0);  // copies > 1: the same basis written `copies` times, `stride` floats apart

// batched square transpose (f32): out[b][j][i] = in[b][i][j], n x n with leading dim ld; batch strides default to n * ld
int transpose_launch(const float* in, float* out, int n, long ld, int batch, hipStream_t s, long stride_in = 0, long stride_out = 0);

// Cholesky of one 64x64 diagonal block per batch item, in place (lower), plus its inverse and inverse^T.
// A: [batch][n][ld]; block k starts at (64k,64k).  Linv/LinvT: [batch][nblk][64][64]
int chol_diag_launch(float* A, long ld, long strideA, float* Linv, float* LinvT, int k, int nblk, int batch,
                     hipStream_t s);

// Round 6 (chol_col.hip): block column k of the left-looking factorisation of the augmented system A [batch][(n + d) x n]
// (ld floats per row): L[R, S_k] for every row block R below the diagonal block incl. the d right-hand-side rows, the inverse
// tables of block k, LT[S_k, R] = L[R, S_k]^T.  The factor's diagonal blocks travel in LT's diagonal blocks until
// chol_col_restore_launch copies them into A (behind the last column).
int chol_col_launch(float* A, long ld, long strideA, float* LT, long strideLT, int n, int d, float* Linv, float* LinvT, int k,
                    int nblk, int batch, unsigned epoch, hipStream_t s);
extern int g_gp_col_leader;  // roma_tuning("gp_col_leader"): leader / follower hand-off inside a column launch on / off
int chol_col_restore_launch(float* A, long ld, long strideA, const float* LT, long strideLT, int n, int batch, hipStream_t s);

// pad the trailing (npad - n) diagonal of a Gram matrix with identity and zero its off-diagonals
int pad_identity_launch(float* A, long ld, long strideA, int n, int npad, int batch, hipStream_t s);

// cls_to_flow_refine (romatch/utils/utils.py:300-322): logits [M, ld] (4096 classes + 1 certainty logit)
// -> flow [M,2] (x,y), cert [M]
int cls_to_flow_launch(const float* logits, long ld, float* flow, float* cert, long M, hipStream_t s);

struct RefinerInputArgs {
  const void* feat = nullptr;  // projected features [nimg, H*W, C] (row stride ldf)
  long ldf = 0;
  const float* flow = nullptr;  // [B, H*W, 2]
  void* d = nullptr;            // [B, H*W, ldd]: [x | x_hat | emb | (corr written by local_corr) | zero pad]
  long ldd = 0;
  const float* emb_w = nullptr;  // [E,2]
  const float* emb_b = nullptr;  // [E]
  int B = 0, H = 0, W = 0, C = 0, E = 0, Kcorr = 0, nimg = 0, shift = 0;
  float disp_scale = 1.f;  // 40/32 * scale_factor
  int dt = 0;
};
int refiner_input_launch(const RefinerInputArgs& a, hipStream_t s);

// depthwise 5x5 (pad 2) + folded BN + ReLU, NHWC.  w: [25][Cp], bias [Cp]
int dwconv5x5_launch(const void* in, void* out, const float* w, const float* bias, int B, int H, int W, int Cp,
                     int dt, hipStream_t s);
// wide 16-bit problems (Cp % 64 == 0, Cp >= 256): the wave-private LDS-DMA ring form (dwconv_ring.hip), bit-identical to the
// kernel above.  0 = launched, 1 = not its problem, < 0 = error.  g_dw_ring: roma_tuning("dw_ring") A/B switch.
int dwconv5x5_ring_try_launch(const void* in, void* out, const float* w, const float* bias, int B, int H, int W, int Cp, int dt,
                              hipStream_t s);
extern int g_dw_ring;
extern int g_pool_proj;  // model.hip: roma_tuning("pool_proj")
extern int g_gp_col;  // model.hip: roma_tuning("gp_col") - left-looking block-column Cholesky (chol_col.hip) on / off

// out_conv (C->3, f32) fused with the flow / certainty update (matcher.py:177-178, 496-506)
int refiner_out_launch(const void* d, long ldd, int dt, const float* w /*[3][Cp]*/, const float* b /*[3]*/,
                       float* flow, float* cert, long M, int Cp, float sx, float sy, hipStream_t s);
// flow[i] += (sx, sy) * delta[i].xy, cert[i] += delta[i].z (delta [M][4] f32: the FINAL refiner blocks, refiner_block.h)
int refiner_apply_delta_launch(const float* delta, float* flow, float* cert, long M, float sx, float sy, hipStream_t s);

// bilinear resize, align_corners=False, channels-last small-channel maps (nc = 1 or 2), f32
int resize_bilinear_launch(const float* in, float* out, int B, int Hin, int Win, int Hout, int Wout, int nc,
                           hipStream_t s);

struct FinalArgs {
  const float* flow = nullptr;    // [b, H, W, 2] finest flow
  const float* cert = nullptr;    // [b, H, W] finest certainty logits
  const float* cert16 = nullptr;  // [b, h16, w16] pass-1 stride-16 certainty logits (attenuation) or null
  float* warp = nullptr;          // [B, H, 2W, 4] (symmetric) or [B, H, W, 4]
  float* certainty = nullptr;     // [B, H, 2W] or [B, H, W]
  int B = 0, H = 0, W = 0, h16 = 0, w16 = 0, symmetric = 1;
};
int final_epilogue_launch(const FinalArgs& a, hipStream_t s);

// out ^= order-independent 64-bit checksum of `bytes` bytes at p (whole 32-bit words); *out must be zeroed by the caller
int checksum_cols_launch(const void* p, long rows, int ld, int c0, int c1, unsigned long long* out, hipStream_t s);
int checksum_launch(const void* p, size_t bytes, unsigned long long* out, hipStream_t s);

}  // namespace roma
