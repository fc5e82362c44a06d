/* libroma_hip - C ABI of the MI355X-native RoMa dense-matching path.
 *
 * The reference (Parskatt/RoMa) is pure Python; the interfaces this library replaces are
 *   - romatch/models/model_zoo/__init__.py:31-93  roma_outdoor / roma_indoor factories
 *                                                 -> roma_create / roma_set_tensor / roma_finalize
 *   - romatch/models/model_zoo/roma_models.py:204 strict load_state_dict       -> roma_set_tensor + roma_finalize
 *   - romatch/models/matcher.py:779-934           RegressionMatcher.match()    -> roma_match
 *   - romatch/models/matcher.py:585-596, 631-670  forward / forward_symmetric / extract_backbone_features -> roma_forward
 *   - romatch/utils/local_correlation.py:22-35    local_corr.local_corr(...) (external fused-local-corr wheel)
 *                                                 -> roma_op_local_corr (plugin signature),
 *                                                    roma_op_local_corr_window (what local_correlation():77-143 needs)
 * plus per-operator entry points so that every kernel can be parity-tested alone.
 *
 * Conventions: plain pointers and sizes only; all tensor pointers are DEVICE pointers unless
 * stated otherwise; `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 * asynchronous on that stream; return 0 on success, negative on error (text: roma_last_error()).
 * One handle per device, one in-flight roma_match per handle (the reference's single-caller model).
 */
#ifndef ROMA_HIP_H
#define ROMA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct roma_model* roma_handle_t;

/* arithmetic / storage type of activations.  The 16-bit format is a property of the library BUILD: libroma_hip.so stores
 * bfloat16 and accepts ROMA_F32 / ROMA_BF16; libroma_hip_f16.so (same sources, -DROMA_H16_F16) stores IEEE binary16 - the
 * reference's default amp_dtype (model_zoo/__init__.py:37) - and accepts ROMA_F32 / ROMA_F16.  The other code is an error
 * (ROMA_ERR_ARG), never a reinterpretation.  roma_h16_format() returns the code of the loaded library. */
enum { ROMA_F32 = 0, ROMA_BF16 = 1, ROMA_F16 = 2,
       /* the precision policy the reference's own timing script runs (tests/test_roma_upsample_inference_time.py:36-45):
        * amp_dtype = bfloat16 reaches DINOv2 only (model_zoo/roma_models.py:183-188); the VGG pyramid, the decoder and the
        * refiners keep binary16 (encoders.py:7, matcher.py:46,341).  Accepted by libroma_hip_f16.so, which loads
        * libroma_hip.so from its own directory and runs DINOv2 there through roma_vit_forward. */
       ROMA_MIXED = 3 };
int roma_h16_format(void);
enum { ROMA_ERR_ARG = -1, ROMA_ERR_HIP = -2, ROMA_ERR_STATE = -3 };

typedef struct {
  int coarse_h, coarse_w;     /* multiples of 14 (DINOv2 patch; roma_models.py:58-59), e.g. 560, 518, 672 */
  int upsample_h, upsample_w; /* any size >= 16 (0 = no upsample pass); used when upsample_preds != 0 */
  int symmetric;              /* matcher.py:801   */
  int upsample_preds;         /* matcher.py:836   */
  int attenuate_cert;         /* matcher.py:839   */
  int precision;              /* ROMA_F32 (exact f32 MFMA; CPU-oracle parity), the library's 16-bit code, or - binary16 build
                                 only - ROMA_MIXED: DINOv2 in bfloat16, everything else in binary16 (the policy of
                                 tests/test_roma_upsample_inference_time.py:36-45 + roma_models.py:183-188) */
  int max_batch;              /* largest number of image pairs per roma_match call                 */
  int device;                 /* HIP device ordinal                                                */
} roma_config_t;

const char* roma_last_error(void);
const char* roma_version(void);
/* ABI stamp of the structures that cross the library boundary BETWEEN the two builds (roma_vit_args_t, roma_vit_block_t:
 * a ROMA_MIXED handle of libroma_hip_f16.so calls roma_vit_forward of libroma_hip.so): ROMA_ABI_VERSION * 100000 +
 * sizeof(roma_vit_args_t).  The sibling is refused unless the stamps are equal. */
#define ROMA_ABI_VERSION 5
int roma_abi_stamp(void);
/* the 16-bit format as THIS library's own internal calls see it (== roma_h16_format() unless another library's symbols
 * interpose: the sibling check of ROMA_MIXED and tests/test_cpu_oracle.py use it) */
int roma_self_check(void);

/* ---- DINOv2 ViT-L/14 as a stand-alone entry (dinov2.py:192-237: patch embed, cls + position embedding, 24 pre-norm
 * blocks, final LayerNorm, patch tokens).  Everything is a DEVICE pointer prepared by the caller: weights in THIS
 * library's 16-bit format (w) or f32 (everything else; LayerScale already folded into proj / fc2), workspace of the
 * sizes given below.  act = ROMA_F32 or this library's 16-bit code.  Used by ROMA_MIXED handles of the other build. */
typedef struct {
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
  const float *ls1, *ls2;                    /* per-channel LayerScale applied in the epilogue, or NULL when folded */
  const void *qkv_w, *proj_w, *fc1_w, *fc2_w; /* [N][ldw], K-contiguous */
  const float *qkv_b, *proj_b, *fc1_b, *fc2_b;
  int qkv_ldw, proj_ldw, fc1_ldw, fc2_ldw;
} roma_vit_block_t;

typedef struct {
  int B, H, W;                      /* B pairs = 2B images: im_a[B,3,H,W], im_b[B,3,H,W] f32 (H, W multiples of 14) */
  int act;                          /* ROMA_F32 or the library's 16-bit code */
  int bf16_residual;                /* 16-bit mode: residual stream in 16 bits (the reference's bf16 backbone) */
  const float *im_a, *im_b;
  const void* patch_w;              /* [1024][patch_ldw], k = c*196 + ky*14 + kx, zero padded */
  const float* patch_b;
  int patch_ldw;
  const float *cls_tok, *pos_emb;   /* [1024], [1 + T][1024] (already resized to the token grid) */
  const roma_vit_block_t* blocks;   /* HOST array */
  int nblocks;
  const float *norm_w, *norm_b;
  /* workspace, T = (H/14)(W/14), rows = 2B (T + 1), Npad = T + 1 rounded up to 128, e = bytes per activation element:
   * col [2B T patch_ldw] e | pt [2B T 1024] f32 | x [rows 1024] f32 | xs [rows 1024] 2 B (16-bit residual only) |
   * ln, ao [rows 1024] e | hid [rows 4096] e | q, k, vt [2B 16 Npad 64] e, zero-initialised once and never written by
   * anyone else (the padding rows must stay zero) */
  void *col, *pt, *x, *xs, *ln, *ao, *hid, *q, *k, *vt;
  void* feat_out;                   /* [2B, T, 1024] patch tokens (x_norm_patchtokens), activation dtype */
} roma_vit_args_t;
int roma_vit_forward(const roma_vit_args_t* a, void* stream);
/* bfloat16 bits -> this library's 16-bit format (the hand-over of a ROMA_MIXED handle), n elements */
int roma_op_convert_from_bf16(const void* in_bf16, void* out_h16, long n, void* stream);

int roma_create(const roma_config_t* cfg, roma_handle_t* out);
/* name: a key of the reference's matcher state-dict, or "dinov2." + a key of the DINOv2 dict.
 * data: HOST pointer to float32 (int64 for num_batches_tracked, ignored). The library copies. */
int roma_set_tensor(roma_handle_t h, const char* name, int ndim, const int64_t* shape, const void* data, int is_int64);
/* strict key/shape check (as load_state_dict strict=True), BN folding, repacking, upload. */
int roma_finalize(roma_handle_t h);
/* mutable attributes of RegressionMatcher (README.md:82-90): "symmetric", "upsample_preds", "attenuate_cert", "debug";
 * tuning: "fuse_refiner_blocks" (default 1; 0 = separate dwconv + GEMM kernels at every scale),
 *         "compose_out_conv" (default 1: the last ConvRefiner block's 1x1 convolution and out_conv - two linear maps with
 *         nothing in between, matcher.py:92-122, 175-178 - are evaluated as the ONE C -> 3 map composed at roma_finalize;
 *         0 = the reference's two steps.  The results differ by rounding only),
 *         "vit_bf16_residual" (bf16 mode only, default 1: DINOv2 residual stream in bf16 like the reference's bf16
 *         backbone, encoders.py; 0 = keep it in f32),
 *         "streams" (1..4, default 2) / "dual_stream" (1 = 2 streams, 0 = 1): run a batch of >= 2 pairs as sub-batches on
 *         several HIP streams (+5 % at batch 8 with 2; side workspaces are allocated on first use; results are
 *         bit-identical to the single-stream schedule - DESIGN.md section 4) */
int roma_set_option(roma_handle_t h, const char* key, int value);
/* "coarse_scale_factor": the displacement-embedding scale of the COARSE pass, sqrt(h_resized * w_resized / 560^2) of the
 * matcher's configured resolution (matcher.py:805) - it differs from the handle's own resolution only when the caller
 * feeds tensors of another size than the matcher was configured for (matcher.py:822-826).  0 = derive from the handle. */
int roma_set_option_f(roma_handle_t h, const char* key, double value);
/* im_*: [B,3,H,W] float32 normalised images (already on the device). *_hr may be NULL when
 * upsample_preds == 0.  warp_out: [B,Ho,2*Wo,4] (symmetric) or [B,Ho,Wo,4]; cert_out: [B,Ho,2*Wo] / [B,Ho,Wo]. */
int roma_match(roma_handle_t h, int B, const float* im_a, const float* im_b, const float* im_a_hr,
               const float* im_b_hr, float* warp_out, float* cert_out, void* stream);
/* ONE decoder pass with its per-scale correspondences exposed: RegressionMatcher.forward / forward_symmetric in eval mode
 * (matcher.py:631-670 -> Decoder.forward, matcher.py:395-527: corresps[s] = {"flow", "certainty"} for s = 16, 8, 4, 2, 1) and
 * extract_backbone_features (matcher.py:585-596).  upsample = 0: the coarse pass over images at the handle's coarse resolution
 * (scales 16 .. 1); upsample = 1: the upsample pass (matcher.py:870-889: scales 8 .. 1, no DINOv2 / GP) over images at the
 * handle's upsample resolution, seeded with batch["corresps"] = (seed_flow, seed_cert) of ANY resolution, which Decoder.forward
 * resizes bilinearly to the stride-8 grid (matcher.py:423-435).  symmetric selects forward_symmetric (decoder batch Bd = 2B:
 * A->B then B->A) or forward (Bd = B).  Outputs are channels-last f32, the caller permutes: flow[i] [Bd, h_s, w_s, 2] (x, y),
 * cert[i] [Bd, h_s, w_s] logits, i = 0 .. 4 for s = 16, 8, 4, 2, 1 (h_16 = H / 14, h_s = H / s otherwise); any pointer may be
 * NULL (that output is skipped; index 0 is ignored in an upsample pass).  feat[i]: the two images' feature pyramid
 * [2B, h_s, w_s, C_s] (C = 1024, 512, 256, 128, 64) in the handle's activation type (f32, or the library's 16-bit format).
 * Runs on the caller's stream, one sub-batch (no stream split); B <= max_batch. */
typedef struct {
  int upsample;
  int symmetric;
  double scale_factor;      /* ConvRefiner displacement scale (matcher.py:805, 877-881); forward()'s default is 1 */
  const float* seed_flow;   /* upsample pass only: [Bd, seed_h, seed_w, 2] f32 */
  const float* seed_cert;   /* [Bd, seed_h, seed_w] f32 logits */
  int seed_h, seed_w;
  float* flow[5];
  float* cert[5];
  void* feat[5];
} roma_forward_args_t;
int roma_forward(roma_handle_t h, int B, const float* im_a, const float* im_b, const roma_forward_args_t* a, void* stream);
/* debug stage capture (enabled by roma_set_option(h,"debug",1)): copies a named intermediate to HOST memory.
 * Returns the number of bytes available when dst == NULL. */
long roma_debug_fetch(roma_handle_t h, const char* name, void* dst_host, long nbytes);
/* debug mode only: replace a named intermediate of the following roma_match calls by the HOST buffer given here
 * (copied; src_host == NULL removes the override).  Stages: "gm_flow16" [b, h16*w16, 2] and "gm_cert16" [b, h16*w16]
 * f32, b = decoder batch - the output of cls_to_flow_refine (utils/utils.py:300-322), whose arg-max is discontinuous:
 * parity tests of the reduced-precision mode inject the oracle's coarse match and bound everything downstream. */
int roma_debug_inject(roma_handle_t h, const char* name, const void* src_host, long nbytes);
/* determinism trace (roma_set_option(h, "trace", 1)): every stage of the following roma_match calls XORs an
 * order-independent 64-bit checksum of its output into a table, one table per sub-batch stream (slot 0 = the caller's
 * stream).  Returns the number of entries of the last call (sums_host == NULL: count only); names_host receives the stage
 * names, newline separated.  tools/stress_streams.py --trace uses it to name the FIRST stage that differs between runs. */
long roma_debug_trace(roma_handle_t h, int slot, unsigned long long* sums_host, long max_entries, char* names_host,
                      long names_bytes);
int roma_destroy(roma_handle_t h);
/* Per-launch HIP-event timing of the dominant kernels (bench.py roofline pass). roma_profile_report writes a JSON
 * object {kernel: {calls,total_ms,work,unit}} (work = algorithmic FLOPs or bytes); returns bytes needed when buf==NULL. */
/* process-wide kernel-selection switches for A/B measurements and tests (not needed for normal use): "gemm8p" 1 / 0 =
 * route the large bf16 GEMMs to the 8-phase kernel or keep them on the one-barrier-per-slab kernel (-1 = environment
 * ROMA_GEMM8P, default on); "gemm_dbg" = experiment bits of the GEMM kernels (-1 = environment ROMA_GEMM_DBG);
 * "lc_mode" = local correlation: 0 tiled LDS form with a per-tile gather work list (default), 1 every tile on the gather
 * list, 2 the per-pixel kernel of round 1 (-1 = environment ROMA_LC_MODE); "conv64" = bit mask of the
 * weight-stationary VGG front-end kernels; "attn_xcd" 1 / 0 = attention work items in per-XCD bands; "attn_exp2" 1 / 0 = tools
 * only: force the 2^x softmax of the 16-bit attention kernel on / off (it assumes q pre-scaled by log2 e); "dw_ring" 0 / 1 / 2 = depthwise 5x5: register-prefetch
 * kernel / wave-private ring kernel for launches >= 64 M elements (default) / ring kernel for every shape it takes;
 * "rb24w" 1 / 0 = C = 24 fused block: wave-private kernel (default) / two-barrier workgroup kernel; "rb144_1b" 1 / 0 = C = 144
 * fused block: one barrier per row (default) / two; "gemm8p_sched" 1 / 0 = K-loop schedule of the 8-phase GEMM: k-half
 * phases (default) / quadrant phases (bit-identical results); "rb_wide" 1 / 0 = the C = 576 ConvRefiner block as ONE fused kernel
 * (measured slower: off) / as dwconv5x5 + 1x1 GEMM (default; ROMA_RB_WIDE=1 enables the fused kernel); "ws1x1" 1 / 0 = the N = K = 576 refiner 1x1 on the
 * weight-stationary kernel (default) / on the 256 x 192 tile kernel (bit-identical results); "gp_col" 1 / 0 = the GP's blocked
 * Cholesky solve left-looking, one launch per block column (default) / the right-looking chain of three launches per column
 * (results agree to f32 rounding); "gp_col_leader" 1 / 0 = inside a column launch one leader workgroup per image factorises
 * the diagonal block and hands its inverse to the row-block workgroups (default) / every workgroup factorises its own copy
 * (bit-identical results; also the time-out path of the hand-off); "gemm8p_maxwg" n = measurement only: cap the persistent grid
 * of the 8-phase GEMM at n workgroups (tools/bench_gemm_burst.py; -1 = one per CU); "gemm8p_walk" g = tile rows per group of its
 * walk through an XCD's band of output tiles (1 = row-major; -1 = row-major below 24 tile columns, 8 from there on; same values in
 * any order: tools/bench_gemm_walk.py).  Every alternative computes the same values (the stencil / block
 * kernels bit for bit); -1 restores the default (or the environment variable of the same name in upper case, ROMA_...). */
int roma_tuning(const char* key, int value);
/* measuring tool (tools/bench_gemm_ablation.py): after a GEMM launched with the "gemm_dbg" trace bit (32768), copies the
 * phase time stamps [workgroup < 16][wave group][K tile < 256][phase] (low 32 bits of s_memtime at each phase's first
 * barrier release) to the host; returns the number of bytes written. */
long roma_debug_gemm_trace(unsigned int* dst_host, long nbytes);
int roma_profile_enable(int on);
long roma_profile_report(char* buf, long nbytes);

/* ---- operator entry points (dt: ROMA_F32 or the library's 16-bit code, ROMA_BF16 / ROMA_F16) ------------ */

/* Drop-in for local_corr.local_corr(feature0[B,HW,C], feature1[B,H,W,C], warp[B,HW,K,2], mode, normalized_coords=True)
 * -> out[B,HW,K]   (local_correlation.py:26-32).  feature0 is expected pre-scaled.  nearest = 0: mode "bilinear"; 1: mode
 * "nearest" (the sample_mode the reference threads through local_correlation.py:19,30,85: the pixel at nearbyint of the
 * un-normalised coordinate, zero outside the image - F.grid_sample(mode="nearest", align_corners=False)). */
int roma_op_local_corr(const void* feature0, const void* feature1, const float* warp, void* out, int B, int H, int W,
                       int C, int K, int nearest, int dt_in, int dt_out, void* stream);
/* Window form: warp is the centre coordinate [B,HW,2]; taps = (2r+1)^2 one-pixel steps; scale multiplies the
 * result (1/sqrt(C) when feature0 is not pre-scaled); out row stride ldo >= K. */
int roma_op_local_corr_window(const void* feature0, const void* feature1, const float* warp, void* out, int B, int H,
                              int W, int C, int radius, float scale, long ldo, int dt_in, int dt_out, void* stream);

/* C[M,N] = act(A[M,K] W[N,K]^T + bias) * scale + res   (batched with element strides; any pointer may be NULL) */
int roma_op_gemm(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, int batch,
                 long sA, long sW, long sC, const float* bias, const float* scale, const float* res, long ldr,
                 int act, float alpha, int dt_in, int dt_out, void* stream);
/* 3x3 conv (pad 1) as implicit GEMM on NHWC input: out[B,H,W,Cout] = relu?(conv(in[B,H,W,Cin], w[Cout][9*Cin]) + bias) */
int roma_op_conv3x3(const void* in, const void* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                    int relu, int dt, void* stream);
/* The same convolution with the weight rows in SLAB-MAJOR K order, w[Cout][k], k = ((ci / 64) * 9 + ky * 3 + kx) * 64 + ci % 64
 * (Cin % 64 == 0) - how the model packs the VGG layers with Cout >= 256 in the 16-bit modes.  Those run on the patch-resident
 * kernel (conv_patch.hip: the activation patch + halo of a 64-channel slab stays in LDS for all nine taps, only the weights
 * stream; roma_tuning("conv_patch", 0) selects the plain implicit GEMM instead - bit-identical results). */
int roma_op_conv3x3_slab(const void* in, const void* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                         int relu, int dt, void* stream);
/* Multi-head attention from a packed qkv activation [B*N, 3*heads*hd] (f32): out[B*N, heads*hd].
 * Workspace q,k,vt must hold B*heads*Npad*hd elements each, Npad = roundup(N,128), zero-initialised. */
int roma_op_attention(const void* q, const void* k, const void* vt, void* out, int B, int heads, int N, int npad, int hd,
                      int dt_in, int dt_out, void* stream);
int roma_op_qkv_scatter_gemm(const void* A, const void* W, const float* bias, void* q, void* k, void* vt, int B, int N,
                             int npad, int heads, int hd, int K, int dt_in, int dt_out, void* stream);
int roma_op_layernorm(const float* x, const float* w, const float* b, void* out, long M, int D, float eps, int dt_out,
                      void* stream);
/* LayerNorm with a typed input (dt_in 0 = f32, 1 = bf16; bf16 input implies bf16 output) - the bf16 residual stream of
 * the DINOv2 blocks in bf16 mode (reference: encoders.py casts the backbone and its input to amp_dtype). */
int roma_op_layernorm_dt(const void* x, int dt_in, const float* w, const float* b, void* out, long M, int D, float eps,
                         int dt_out, void* stream);
/* bf16 GEMM with a bf16 residual: C = bf16( res + scale * (A W^T + bias) ), C may alias res (the in-place residual
 * update x += ls * linear(y) of a transformer block, dinov2.py NestedTensorBlock).  N, ldc, ldr multiples of 8. */
int roma_op_gemm_res_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K,
                          const float* bias, const float* scale, const void* res, long ldr, void* stream);
/* Batched SPD solve  (A + 0 ) X = F  via blocked Cholesky: A [batch,n,n] f32 (destroyed), Ft [batch, d, n] = F^T,
 * overwritten by X^T.  Workspaces: LT [batch,n,n], Linv/LinvT [batch, n/64, 64, 64].  n multiple of 64.
 * Augmented layout (what the GP uses): Ft == A + n * n, i.e. ONE (n + d) x n matrix per item with the right-hand sides right
 * behind A (items (n + d) * n floats apart when batch > 1) - the forward substitution then runs inside the factorisation,
 * one launch per 64-column block (chol_col.hip; roma_tuning("gp_col", 0) selects the right-looking launch chain instead). */
int roma_op_cholesky_solve_t(float* A, float* Ft, float* LT, float* Linv, float* LinvT, int n, int d, int batch,
                             void* stream);
/* GP.forward (matcher.py:291-323) with the cosine kernel (matcher.py:191-200, T = 0.2) and sigma_noise = 0.1:
 *   mu[i] = K(x_i, y_i) (K(y_i, y_i) + 0.1 I)^-1 cos(8 pi (pos_w . grid + pos_b))
 * x, y: [b, h*w, 512] channels-last stride-16 features (dt), pos_w [512,2], pos_b [512] f32 (decoder.gps.16.pos_conv),
 * mu: [b, h*w, 512] f32.  Always evaluated in f32 (Gram matrices via exact-f32 MFMA when dt = ROMA_F32, blocked
 * Cholesky, two triangular solves); scratch is stream-ordered. */
int roma_op_gp(const void* x, const void* y, const float* pos_w, const float* pos_b, float* mu, int b, int h, int w, int dt,
               void* stream);
int roma_op_cls_to_flow(const float* logits, long ld, float* flow, float* cert, long M, void* stream);
int roma_op_resize_bilinear(const float* in, float* out, int B, int Hin, int Win, int Hout, int Wout, int nc,
                            void* stream);
/* The ConvRefiner input writer = F.grid_sample warp + concat (matcher.py:132-148, 166), one pass:
 *   d[b,p,:] = [ x[b,p,0:C] | bilinear_zeropad(y[(b+shift) % nimg], flow[b,p]) (align_corners=False) |
 *                emb_w . (disp_scale * (flow[b,p] - grid[p])) + emb_b (E values) | Kcorr columns left untouched (local
 *                correlation writes them) | zeros up to ldd ]
 * feat: projected features of all nimg images [nimg, H*W, ldf] (x = image b, y = image (b+shift) % nimg), flow [B,H*W,2]
 * f32 normalised (x,y), emb_w [E,2], emb_b [E] f32, disp_scale = 40/32 * scale_factor. */
int roma_op_refiner_input(const void* feat, long ldf, const float* flow, void* d, long ldd, const float* emb_w,
                          const float* emb_b, int B, int H, int W, int C, int E, int Kcorr, int nimg, int shift,
                          float disp_scale, int dt, void* stream);
int roma_op_dwconv5x5(const void* in, void* out, const float* w, const float* bias, int B, int H, int W, int Cp, int dt,
                      void* stream);
/* One fused ConvRefiner block (matcher.py:88-117 create_block): out = conv1x1(relu(bn(dwconv5x5(in)))), BN folded into
 * dw_w/dw_b.  bf16 only, Cp in {24, 144} (narrow scales) or 576 (round 5: the stride-4 scale, all output channels per
 * workgroup); in/out [B,H,W,Cp] must not alias; pw bf16 [Cp][Cp], pw_b f32 [Cp]. */
int roma_op_refiner_block(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw,
                          const float* pw_b, int B, int H, int W, int Cp, int dt, void* stream);
/* The LAST block of a narrow ConvRefiner with its 1x1 composed with out_conv (two linear maps back to back, matcher.py:92-122,
 * 175-178; "compose_out_conv"): delta[pixel] = {d flow x, d flow y, d certainty, 0} (f32 [B*H*W][4]) instead of a block output.
 * pw_final: 16-bit [8][Cp], rows 0-2 = 16-bit head and rows 4-6 = 16-bit remainder of the composed [3][Cp] weights (rows 3, 7
 * zero); bias_final f32 [Cp] (composed bias in [0, 3), zeros behind).  bf16 / f16 only, Cp in {24, 144}. */
int roma_op_refiner_block_final(const void* in, float* delta, const float* dw_w, const float* dw_b, const void* pw_final,
                                const float* bias_final, int B, int H, int W, int Cp, int dt, void* stream);
/* flow[i] += (sx, sy) * delta[i].xy, cert[i] += delta[i].z (the deltas above; flow [M][2], cert [M], f32) */
int roma_op_refiner_apply_delta(const float* delta, float* flow, float* cert, long M, float sx, float sy, void* stream);
/* Gaussian KDE of sampled matches (romatch/utils/kde.py:4-12; RegressionMatcher.sample, matcher.py:598-629):
 * density[i] = sum_j exp(-|x_i - x_{j*down}|^2 / (2 std^2)), x: DEVICE [n,4] f32.  half_inputs != 0 rounds the
 * coordinates to fp16 first (the reference's x.half()); accumulation is f32. */
int roma_op_kde(const float* x, long n, int down, float std, int half_inputs, float* density, void* stream);
/* RegressionMatcher.match_keypoints (matcher.py:732-773), all pointers DEVICE:
 * sample_warp_at: xa_to_b[i] = bilinear(warp[..., 2:4], xa[i]), cert_a[i] = bilinear(cert, xa[i])  (zeros padding,
 *   align_corners=False); warp [H,W,4] f32, cert [H,W] f32, xa [n,2] normalised (x,y).
 * mutual_nn: match_b[i] = j if b[j] is the nearest neighbour of a[i], a[i] is at the column-minimum distance of
 *   b[j], cert_a[i] > cert_th (cert_a may be NULL) and |a[i]-b[j]| < max_dist; else -1.  Row ties resolve to the
 *   lowest j.  ws_a / ws_b: 8*na / 8*nb byte workspaces.
 * mutual_nn_count / mutual_nn_fill: the tie-complete form, i.e. torch.nonzero of the reference's mask
 *   (D == row min) * (D == column min) * (cert > th) * (D < max_dist) (matcher.py:756-762) with EVERY tied pair (duplicate
 *   keypoints), row-major.  count runs the two nearest-neighbour passes and writes offsets[0..na] (int64): the exclusive
 *   prefix sums of the per-row match counts, offsets[na] = number of pairs; the caller reads that one value, allocates
 *   pairs[n][2] (int64: index into a, index into b) and calls fill with the same arguments and workspaces. */
int roma_op_sample_warp_at(const float* warp, const float* cert, int H, int W, const float* xa, long n, float* xa_to_b,
                           float* cert_a, void* stream);
int roma_op_mutual_nn(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                      int* match_b, void* ws_a, void* ws_b, void* stream);
int roma_op_mutual_nn_count(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                            void* ws_a, void* ws_b, long long* offsets, void* stream);
int roma_op_mutual_nn_fill(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                           const void* ws_a, const void* ws_b, long long* offsets, long long* pairs, void* stream);
/* torch.multinomial(weights, k, replacement=False) of RegressionMatcher.sample (matcher.py:615-627): k distinct int64
 * indices, drawn with probability proportional to the non-negative f32 weights [n] (exponential race + radix select, no
 * sort; reproducible from `seed`), returned in DRAW order (ascending race key) like torch.multinomial, so a prefix of the
 * result is itself a valid smaller sample.  If fewer than k weights are positive, zero-weight entries complete the sample
 * (last), as on torch's GPU path.  workspace: device memory of roma_op_multinomial_workspace(n, k) bytes. */
long roma_op_multinomial_workspace(long n, long k);
int roma_op_multinomial(const float* weights, long n, long k, unsigned long long seed, long long* out_indices, void* workspace,
                        long workspace_bytes, void* stream);
/* ---- Tiny RoMa (romatch/models/tiny.py), matcher side; the XFeat backbone is the caller's (model_zoo/__init__.py:24-27).
 * All tensors f32, channels-last unless noted.  corr_volume (tiny.py:182-196) = roma_op_gemm with A = feats of image B
 * [H1*W1, C], W = feats of image A [H0*W0, C], alpha = 1/sqrt(C), batch = pairs: cv [B, H1*W1, H0*W0]. */
int roma_op_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, void* stream);
/* pos_embed (tiny.py:114-142): out [B, H0*W0, 2].  exact_softmax = 0: the inference path, soft arg-max over the
 * 4x-subsampled correlation column plus the arg-max position (H1, W1 multiples of 4); 1: the exact_softmax=True branch
 * (tiny.py:139-141), the softmax over all H1 x W1 positions. */
int roma_op_tiny_pos_embed(const float* corr_volume, float* out, int B, int H1, int W1, int H0, int W0, int exact_softmax,
                           void* stream);
/* TinyRoMa.forward_single (tiny.py:81-99) - the caller's XFeat network, replayed layer by layer, channels-last f32:
 * gray_instnorm: out [B,H,W,1] = InstanceNorm2d(1)(mean over the C channels of in [B,H,W,C]) (no affine, biased variance);
 * conv2d_nhwc: out [B,Ho,Wo,Cout] = act(conv(in [B,H,W,Cin], w [K*K*Cin][Cout] (tap-major, BatchNorm folded)) + bias) + res;
 *   K 1 or 3, stride 1 or 2, padding 0 or 1, Cout % 4 == 0; bias, res may be NULL;  avgpool_nhwc: AvgPool2d(k, k);
 * add3: out = a + b (+ c, may be NULL).  The bilinear resizes are roma_op_resize_bilinear. */
int roma_op_gray_instnorm(const float* in, float* out, int B, int H, int W, int C, float eps, void* stream);
int roma_op_conv2d_nhwc(const float* in, const float* w, const float* bias, const float* res, float* out, int B, int H, int W,
                        int Cin, int Cout, int K, int stride, int pad, int relu, void* stream);
int roma_op_avgpool_nhwc(const float* in, float* out, int B, int H, int W, int C, int k, void* stream);
int roma_op_add3(const float* a, const float* b, const float* c, float* out, long n, void* stream);
/* d[B,H,W,Cp] = cat(f0 [B,H,W,C], grid_sample(f1 [B,H1,W1,C], warp[..., 0:2]), warp[..., 0:2], zero pad)  (tiny.py:290-291,
 * 298-299; bilinear, zeros padding, align_corners=False); warp has warp_channels >= 2 channels per pixel. */
int roma_op_tiny_matcher_input(const float* f0, const float* f1, const float* warp, int warp_channels, float* d, int B, int H,
                               int W, int H1, int W1, int C, int Cp, void* stream);
/* out[p, 0:3] = base[p, 0:base_channels] (third channel 0 when base has 2) + delta[p, 0:3] * (sx, sy, 1)  (tiny.py:289-300) */
int roma_op_tiny_update(const float* base, int base_channels, const float* delta, long ldd, float sx, float sy, float* out,
                        long npix, void* stream);
/* warp [B,H,W,4] = (grid, matches[..., 0:2]), certainty [B,H,W] = sigmoid(matches[..., 2])  (tiny.py:226-238) */
int roma_op_tiny_final(const float* matches, float* warp, float* certainty, int B, int H, int W, void* stream);
/* RegressionMatcher.visualize_warp (matcher.py:936-986): out[c,y,x] = certainty * grid_sample(image, warp) + (1 - certainty)
 * (bilinear, zeros padding, align_corners=False; white background).  warp [H, W2, 4] f32 with W2 = 2W (symmetric: left
 * half samples im_b at warp[..., 2:4], right half samples im_a at warp[..., 0:2], matcher.py:967-975) or W2 = W (im_a may
 * be NULL); certainty [H, W2]; images [3, im_h, im_w] f32; out [3, H, W2] f32. */
int roma_op_visualize_warp(const float* warp, const float* certainty, const float* im_a, const float* im_b, int H, int W,
                           int symmetric, int im_h, int im_w, float* out, void* stream);
/* conf_from_fb_consistency (matcher.py:672-699): out[b,y,x] = 1 if the backward flow sampled (bilinear, zeros padding,
 * align_corners=False) at the forward flow's target returns to within th_n of pixel (x,y)'s own normalised
 * coordinate, else 0.  flows DEVICE [B,H,W,2] f32, out [B,H,W] f32; th_n = 2*th / max(H,W). */
int roma_op_fb_consistency(const float* flow_fwd, const float* flow_bwd, int B, int H, int W, float th_n, float* out,
                           void* stream);
int roma_op_maxpool2x2(const void* in, void* out, int B, int H, int W, int C, int dt, void* stream);
/* MaxPool2d(2) + the proj head of a VGG pyramid level in one pass over the un-pooled map (encoders.py:17-27,
 * roma_models.py:156-160): in [B,H,W,C] 16-bit, C = 64 (N <= 32) or 128 (N <= 64) -> pooled [B,H/2,W/2,C] and
 * pf [B,H*W,ldf] = in . pw^T + pb (pw [N][ldw] 16-bit, pb f32 [N], columns N .. ldf zero).  Bit-identical to
 * roma_op_maxpool2x2 + roma_op_gemm; roma_tuning("pool_proj", 0) makes the model use those two instead. */
int roma_op_pool_proj(const void* in, void* pooled, void* pf, const void* pw, long ldw, const float* pb, int N, int ldf, int B, int H,
                      int W, int C, int dt, void* stream);
/* ConvRefiner out_conv fused with the flow / certainty update (matcher.py:177-178, 496-506):
 *   o = d[m, 0:Cp] . w[0:3, 0:Cp]^T + b;  flow[m] += (sx * o0, sy * o1);  cert[m] += o2        (f32 accumulate)
 * d DEVICE [M, ldd] in dt (f32 / bf16; channels Cp..ldd ignored), w DEVICE f32 [3][Cp], b f32 [3], flow f32 [M,2], cert f32 [M]. */
int roma_op_refiner_out(const void* d, long ldd, int dt, const float* w, const float* b, float* flow, float* cert, long M,
                        int Cp, float sx, float sy, void* stream);
int roma_op_conv3x3_c3(const float* img, const float* w, const float* bias, void* out, int B, int H, int W, int dt_out,
                       void* stream);
/* First VGG19-BN layer of the bf16 path (encoders.py:17-27, features[0..2] with the BatchNorm folded): img DEVICE f32
 * [B,3,H,W] -> out DEVICE bf16 [B,H,W,64] = ReLU(conv3x3(bf16(img), w, pad 1) + bias), products exact, f32 accumulate.
 * w DEVICE bf16 [64][32] with column k = ci*9 + ky*3 + kx (columns 27..31 zero), bias DEVICE f32 [64]. */
int roma_op_conv3x3_c3_bf16(const float* img, const void* w, const float* bias, void* out, int B, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif
