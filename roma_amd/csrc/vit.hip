// DINOv2 ViT-L/14 forward and the shared pre-norm transformer block (vit.h).  Schedule only: every kernel is one of
// gemm.hip / gemm8p.hip (linears with fused bias / GELU / residual / QKV scatter), attention.hip, elementwise.hip.
#include "vit.h"

#include <algorithm>

#include "attention.h"
#include "elementwise.h"
#include "gemm.h"

namespace roma {

// roma_self_check (api.hip): the format code as a CROSS-translation-unit internal call sees it
int internal_h16_code() {
#ifdef ROMA_H16_F16
  return ROMA_F16;
#else
  return ROMA_BF16;
#endif
}


#define VRUN(expr)          \
  do {                      \
    int _rc = (expr);       \
    if (_rc) return _rc;    \
  } while (0)

// layers/block.py:82-107 (DINOv2, LayerScale) and :36-65 (decoder blocks, no LayerScale):
//   x += ls1 * proj(SDPA(qkv(LN1(x))));  x += ls2 * fc2(GELU(fc1(LN2(x))))
int vit_block_run(const roma_vit_block_t& w, void* x, int x_dt, long rows, int Bn, int N, int npad, int heads, int hd,
                  float eps, int act_dt, const VitScratch& s, hipStream_t st) {
  auto residual_gemm = [&](GemmArgs& g) -> int {
    g.C = x; g.ldc = 1024; g.ldr = 1024;
    if (x_dt == DT_BF16) { g.out_dt = DT_BF16; g.res_bf16 = x; }
    else { g.out_dt = DT_F32; g.res = (const float*)x; }
    return gemm_launch(g, st);
  };
  VRUN(layernorm_launch_dt(x, x_dt, w.ln1_w, w.ln1_b, s.ln, rows, 1024, eps, act_dt, st));
  {
    GemmArgs g;
    g.A = s.ln; g.lda = 1024; g.W = w.qkv_w; g.ldw = w.qkv_ldw; g.M = (int)rows; g.N = 3072; g.K = 1024;
    g.in_dt = act_dt; g.out_dt = act_dt; g.bias = w.qkv_b; g.mode = EPI_QKV;
    g.q = s.q; g.k = s.k; g.vt = s.vt; g.heads = heads; g.hd = hd; g.ntok = N; g.npad = npad;
    // 16-bit mode: fold log2(e) into the query scale so the softmax is a bare v_exp_f32 (2^x) per element
    g.qscale = (act_dt == DT_BF16 ? 1.4426950408889634f : 1.0f) / sqrtf((float)hd);
    VRUN(gemm_launch(g, st));
  }
  {
    AttnArgs a;
    a.q = s.q; a.k = s.k; a.vt = s.vt; a.out = s.ao; a.B = Bn; a.heads = heads; a.N = N; a.npad = npad; a.hd = hd;
    a.ldo = 1024; a.in_dt = act_dt; a.out_dt = act_dt; a.exp2_domain = act_dt == DT_BF16 ? 1 : 0;
    VRUN(attention_launch(a, st));
  }
  {
    GemmArgs g;
    g.A = s.ao; g.lda = 1024; g.W = w.proj_w; g.ldw = w.proj_ldw; g.M = (int)rows; g.N = 1024; g.K = 1024;
    g.in_dt = act_dt; g.bias = w.proj_b; g.scale = w.ls1;
    VRUN(residual_gemm(g));
  }
  VRUN(layernorm_launch_dt(x, x_dt, w.ln2_w, w.ln2_b, s.ln, rows, 1024, eps, act_dt, st));
  {
    GemmArgs g;
    g.A = s.ln; g.lda = 1024; g.W = w.fc1_w; g.ldw = w.fc1_ldw; g.C = s.hid; g.ldc = 4096; g.M = (int)rows; g.N = 4096; g.K = 1024;
    g.in_dt = act_dt; g.out_dt = act_dt; g.bias = w.fc1_b; g.act = ACT_GELU;
    VRUN(gemm_launch(g, st));
  }
  {
    GemmArgs g;
    g.A = s.hid; g.lda = 4096; g.W = w.fc2_w; g.ldw = w.fc2_ldw; g.M = (int)rows; g.N = 1024; g.K = 4096;
    g.in_dt = act_dt; g.bias = w.fc2_b; g.scale = w.ls2;
    VRUN(residual_gemm(g));
  }
  return 0;
}

// dinov2.py:192-237 (prepare_tokens_with_masks + blocks + norm) and encoders.py:64-65 (x_norm_patchtokens)
int vit_forward(const roma_vit_args_t& a, hipStream_t st) {
  ROMA_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.H % 14 == 0 && a.W % 14 == 0, "vit_forward: bad image size");
  ROMA_REQUIRE(a.blocks && a.nblocks > 0 && a.im_a && a.im_b && a.feat_out, "vit_forward: null argument");
  const int act_dt = a.act == ROMA_F32 ? DT_F32 : DT_BF16;
  const size_t esz = act_dt == DT_F32 ? 4 : 2;
  const int nimg = 2 * a.B, th = a.H / 14, tw = a.W / 14, T = th * tw;
  const int Nd = T + 1, Npd = (int)round_up(Nd, 128);
  const long rows_d = (long)nimg * Nd;
  auto off = [&](void* p, long elems) -> void* { return static_cast<char*>(p) + elems * (long)esz; };
  VRUN(im2col_patch14_launch(a.im_a, a.col, a.B, a.H, a.W, a.patch_ldw, act_dt, st));
  VRUN(im2col_patch14_launch(a.im_b, off(a.col, (long)a.B * T * a.patch_ldw), a.B, a.H, a.W, a.patch_ldw, act_dt, st));
  {
    GemmArgs g;
    g.A = a.col; g.lda = a.patch_ldw; g.W = a.patch_w; g.ldw = a.patch_ldw; g.C = a.pt; g.ldc = 1024;
    g.M = nimg * T; g.N = 1024; g.K = a.patch_ldw; g.k_alg = 588; g.in_dt = act_dt; g.out_dt = DT_F32; g.bias = a.patch_b;
    VRUN(gemm_launch(g, st));
  }
  VRUN(assemble_tokens_launch((const float*)a.pt, a.cls_tok, a.pos_emb, (float*)a.x, nimg, T, 1024, st));
  void* xs = a.x;
  int x_dt = DT_F32;
  if (act_dt == DT_BF16 && a.bf16_residual) {  // 16-bit residual stream: the reference's bf16 backbone adds in bf16 too
    ROMA_REQUIRE(a.xs, "vit_forward: 16-bit residual stream needs the xs workspace");
    xs = a.xs;
    x_dt = DT_BF16;
    VRUN(copy2d_launch(a.x, 1024, DT_F32, xs, 1024, DT_BF16, rows_d, 1024, st));
  }
  VitScratch s;
  s.ln = a.ln; s.ao = a.ao; s.hid = a.hid; s.q = a.q; s.k = a.k; s.vt = a.vt;
  for (int i = 0; i < a.nblocks; ++i) VRUN(vit_block_run(a.blocks[i], xs, x_dt, rows_d, nimg, Nd, Npd, 16, 64, 1e-6f, act_dt, s, st));
  VRUN(layernorm_launch_dt(xs, x_dt, a.norm_w, a.norm_b, a.ln, rows_d, 1024, 1e-6f, act_dt, st));
  for (int i = 0; i < nimg; ++i)  // drop the cls token: x_norm_patchtokens
    VRUN(copy2d_launch(off(a.ln, ((long)i * Nd + 1) * 1024), 1024, act_dt, off(a.feat_out, (long)i * T * 1024), 1024, act_dt, T, 1024, st));
  return 0;
}
#undef VRUN

}  // namespace roma
