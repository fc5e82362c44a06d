// extern "C" boundary of libroma_hip (declarations + reference citations: include/roma_hip.h)
#include <algorithm>
#include <mutex>
#include <map>
#include <vector>
#include <dlfcn.h>
#include <string.h>

#include "../../include/roma_hip.h"
#include "attention.h"
#include "conv64.h"
#include "conv_patch.h"
#include "elementwise.h"
#include "gemm.h"
#include "local_corr.h"
#include "pool_proj.h"
#include "model.h"
#include "refiner_block.h"
#include "kde.h"
#include "keypoints.h"
#include "tiny.h"
#include "vit.h"
#include "sampling.h"

namespace roma {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace roma

namespace roma {
struct ProfRec { std::string name, unit; double work; hipEvent_t e0, e1; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
bool prof_enabled() { return g_prof_on; }
void prof_begin(const char* kernel, double work, const char* unit, hipStream_t s) {
  ProfRec r;
  r.name = kernel; r.unit = unit; r.work = work;
  (void)hipEventCreate(&r.e0);
  (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, s);
  g_prof.push_back(r);
}
void prof_end(hipStream_t s) { (void)hipEventRecord(g_prof.back().e1, s); }
}  // namespace roma

struct roma_model {
  roma::Model m;
};

namespace roma {
// ROMA_MIXED: the sibling bfloat16 library (once a mixed handle has loaded it) keeps its own switches and its own launch
// profile; tuning calls and the profile of THIS library are forwarded / merged so that callers see one library.
void* g_peer_lib = nullptr;  // set by Model::load_peer (model.hip)
int g_mixed_handles = 0;      // live ROMA_MIXED handles (model.hip): the sibling follows roma_tuning / roma_profile_* only while > 0
template <typename F> static F peer_sym(const char* name) {
  return g_peer_lib ? reinterpret_cast<F>(dlsym(g_peer_lib, name)) : nullptr;
}
}  // namespace roma

using namespace roma;

static inline hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }
// public dtype code -> internal one.  The 16-bit storage format is a property of the BUILD (common.h): this library accepts
// ROMA_F32 and its own 16-bit code only; the other one fails loudly instead of reinterpreting the bits.
#ifdef ROMA_H16_F16
static constexpr int ROMA_H16_CODE = ROMA_F16, ROMA_OTHER16_CODE = ROMA_BF16;
#else
static constexpr int ROMA_H16_CODE = ROMA_BF16, ROMA_OTHER16_CODE = ROMA_F16;
#endif
static inline int dt_code(int dt) {
  if (dt == ROMA_F32) return DT_F32;
  if (dt == ROMA_H16_CODE) return DT_BF16;
  set_error(dt == ROMA_OTHER16_CODE
                ? std::string("this library stores ") + ROMA_H16_NAME + ": load " +
                      (ROMA_H16_CODE == ROMA_BF16 ? "libroma_hip_f16.so for ROMA_F16" : "libroma_hip.so for ROMA_BF16")
                : std::string("bad dtype code"));
  return -1;
}
#define DT(dt)                           \
  ({                                     \
    const int _d = dt_code(dt);          \
    if (_d < 0) return ROMA_ERR_ARG;     \
    _d;                                  \
  })

extern "C" {

const char* roma_last_error(void) { return g_err.c_str(); }
const char* roma_version(void) { return "roma_hip 0.3 (gfx950, 16-bit storage = " ROMA_H16_NAME ")"; }
int roma_h16_format(void) { return ROMA_H16_CODE; }
int roma_abi_stamp(void) { return ROMA_ABI_VERSION * 100000 + (int)sizeof(roma_vit_args_t); }
// roma::internal_h16_code lives in ANOTHER translation unit (vit.hip) on purpose: a call inside one TU binds locally under
// clang's default -fno-semantic-interposition, a cross-TU call goes through the PLT unless the library is linked -Bsymbolic
int roma_self_check(void) { return roma::internal_h16_code(); }

int roma_create(const roma_config_t* cfg, roma_handle_t* out) {
  ROMA_REQUIRE(cfg && out, "roma_create: null argument");
  ROMA_REQUIRE(cfg->coarse_h > 0 && cfg->coarse_w > 0 && cfg->coarse_h % 14 == 0 && cfg->coarse_w % 14 == 0,
               "Needs to be multiple of 14 for backbone");  // roma_models.py:58-59
  // no multiple-of-8 requirement (the reference has none either, e.g. 518 / 574 / 602): the VGG pyramid takes the
  // floor-divided sizes of its 2x2 max-pools and every decoder resize goes to those sizes (model.hip)
  ROMA_REQUIRE(cfg->coarse_h >= 16 && cfg->coarse_w >= 16, "roma_create: coarse resolution too small for the stride-16 pyramid");
  ROMA_REQUIRE(cfg->upsample_h >= 0 && cfg->upsample_w >= 0 && (cfg->upsample_h == 0 || (cfg->upsample_h >= 16 && cfg->upsample_w >= 16)),
               "roma_create: bad upsample resolution");
  ROMA_REQUIRE(cfg->max_batch >= 1, "roma_create: max_batch must be >= 1");
  if (cfg->precision == ROMA_MIXED) {  // bf16 DINOv2 (sibling library) + binary16 elsewhere: a mode of the binary16 build
    ROMA_REQUIRE(ROMA_H16_CODE == ROMA_F16, "ROMA_MIXED: load libroma_hip_f16.so (it runs DINOv2 through libroma_hip.so)");
  } else if (dt_code(cfg->precision) < 0) {
    return ROMA_ERR_ARG;  // ROMA_F32 or this build's 16-bit format
  }
  int ndev = 0;
  ROMA_CHECK_HIP(hipGetDeviceCount(&ndev));
  ROMA_REQUIRE(cfg->device >= 0 && cfg->device < ndev, "roma_create: no such HIP device (the HIP path has no CPU fallback)");
  roma_model* h = new roma_model();
  h->m.cfg = *cfg;
  *out = h;
  return 0;
}

int roma_set_tensor(roma_handle_t h, const char* name, int ndim, const int64_t* shape, const void* data, int is_int64) {
  ROMA_REQUIRE(h, "roma_set_tensor: null handle");
  return h->m.set_tensor(name, ndim, shape, data, is_int64);
}

int roma_finalize(roma_handle_t h) {
  ROMA_REQUIRE(h, "roma_finalize: null handle");
  return h->m.finalize();
}

int roma_set_option(roma_handle_t h, const char* key, int value) {
  ROMA_REQUIRE(h && key, "roma_set_option: null argument");
  const std::string k(key);
  if (k == "symmetric") h->m.cfg.symmetric = value ? 1 : 0;
  else if (k == "upsample_preds") {
    ROMA_REQUIRE(!value || h->m.cfg.upsample_h > 0, "roma_set_option: handle was created without an upsample resolution");
    h->m.cfg.upsample_preds = value ? 1 : 0;
  } else if (k == "attenuate_cert") h->m.cfg.attenuate_cert = value ? 1 : 0;
  else if (k == "debug") h->m.debug = value != 0;
  else if (k == "fuse_refiner_blocks") h->m.fuse_refiner_blocks = value != 0;
  else if (k == "compose_out_conv") h->m.compose_out_conv = value != 0;
  else if (k == "vit_bf16_residual") h->m.vit_bf16_residual = value != 0;
  else if (k == "dual_stream") h->m.n_streams = value != 0 ? 2 : 1;
  else if (k == "trace") h->m.trace_on = value != 0;
  else if (k == "streams") {
    ROMA_REQUIRE(value >= 1 && value <= Model::MAX_STREAMS, "roma_set_option: streams must be 1..4");
    h->m.n_streams = value;
  }
  else {
    set_error("roma_set_option: unknown key " + k);
    return ROMA_ERR_ARG;
  }
  return 0;
}

int roma_set_option_f(roma_handle_t h, const char* key, double value) {
  ROMA_REQUIRE(h && key, "roma_set_option_f: null argument");
  const std::string k(key);
  if (k == "coarse_scale_factor") {
    ROMA_REQUIRE(value >= 0.0, "roma_set_option_f: coarse_scale_factor must be >= 0 (0 = derive from the handle's resolution)");
    h->m.coarse_scale_factor = value;
  } else {
    set_error("roma_set_option_f: unknown key " + k);
    return ROMA_ERR_ARG;
  }
  return 0;
}

int roma_match(roma_handle_t h, int B, const float* im_a, const float* im_b, const float* im_a_hr, const float* im_b_hr,
               float* warp_out, float* cert_out, void* stream) {
  ROMA_REQUIRE(h, "roma_match: null handle");
  return h->m.match(B, im_a, im_b, im_a_hr, im_b_hr, warp_out, cert_out, S(stream));
}

int roma_forward(roma_handle_t h, int B, const float* im_a, const float* im_b, const roma_forward_args_t* a, void* stream) {
  ROMA_REQUIRE(h, "roma_forward: null handle");
  return h->m.forward(B, im_a, im_b, a, S(stream));
}

long roma_debug_fetch(roma_handle_t h, const char* name, void* dst_host, long nbytes) {
  if (!h || !name) return ROMA_ERR_ARG;
  auto it = h->m.dbg.find(name);
  if (it == h->m.dbg.end()) {
    set_error(std::string("roma_debug_fetch: no such stage: ") + name);
    return ROMA_ERR_ARG;
  }
  if (!dst_host) return (long)it->second.second;
  if ((size_t)nbytes < it->second.second) {
    set_error("roma_debug_fetch: destination too small");
    return ROMA_ERR_ARG;
  }
  if (hipSetDevice(h->m.cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return ROMA_ERR_HIP;
  if (hipMemcpy(dst_host, it->second.first, it->second.second, hipMemcpyDeviceToHost) != hipSuccess) return ROMA_ERR_HIP;
  return (long)it->second.second;
}

int roma_debug_inject(roma_handle_t h, const char* name, const void* src_host, long nbytes) {
  ROMA_REQUIRE(h, "roma_debug_inject: null handle");
  return h->m.debug_inject(name, src_host, nbytes > 0 ? (size_t)nbytes : 0);
}

long roma_debug_trace(roma_handle_t h, int slot, unsigned long long* sums_host, long max_entries, char* names_host,
                      long names_bytes) {
  if (!h || slot < 0 || slot >= Model::MAX_STREAMS_DECL) return ROMA_ERR_ARG;
  const long n = h->m.trace_n[slot];
  if (!sums_host) return n;
  if (max_entries < n || !h->m.trace_dev[slot]) {
    set_error("roma_debug_trace: destination too small or tracing was never enabled");
    return ROMA_ERR_ARG;
  }
  if (hipSetDevice(h->m.cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return ROMA_ERR_HIP;
  if (hipMemcpy(sums_host, h->m.trace_dev[slot], (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
    return ROMA_ERR_HIP;
  if (names_host && names_bytes > 0) {
    std::string all;
    for (long i = 0; i < n; ++i) all += h->m.trace_names[slot][(size_t)i] + "\n";
    const size_t c = std::min<size_t>(all.size(), (size_t)names_bytes - 1);
    memcpy(names_host, all.data(), c);
    names_host[c] = 0;
  }
  return n;
}

int roma_destroy(roma_handle_t h) {
  delete h;
  return 0;
}

int roma_tuning(const char* key, int value) {
  ROMA_REQUIRE(key, "roma_tuning: null key");
  // ROMA_MIXED handles run DINOv2 in the bfloat16 sibling, so its switches follow - but only while such a handle is alive:
  // the sibling is the same process-wide instance a pure-bf16 matcher uses (roma_amd loads it too), and a tuning call on
  // the binary16 library must not silently retune an unrelated bf16 matcher once the last mixed handle is gone.
  if (g_mixed_handles > 0)
    if (auto f = peer_sym<int (*)(const char*, int)>("roma_tuning"))
      if (int rc = f(key, value)) {
        set_error(std::string("roma_tuning: the bfloat16 sibling library refused key ") + key);
        return rc;
      }
  const std::string k(key);
  if (k == "gemm8p") g_gemm_tuning[0] = value;
  else if (k == "gemm_dbg") g_gemm_tuning[1] = value;
  else if (k == "gemm8p_walk") g_gemm8p_walk = value;
  else if (k == "gemm8p_sched") g_gemm8p_sched = value;
  else if (k == "gemm8p_maxwg") g_gemm8p_maxwg = value;
  else if (k == "ws1x1") g_ws1x1_mode = value;
  else if (k == "lc_mode") g_lc_mode = value;
  else if (k == "lc_bin") g_lc_bin = value;
  else if (k == "conv64") g_conv64_mode = value;
  else if (k == "conv_patch") g_conv_patch = value;
  else if (k == "attn_xcd") g_attn_xcd_map = value;
  else if (k == "attn_exp2") g_attn_exp2 = value;
  else if (k == "rb24w") g_rb24_wave = value;
  else if (k == "rb144_1b") g_rb144_1b = value;
  else if (k == "rb_wide") g_rb_wide = value;
  else if (k == "dw_ring") g_dw_ring = value;
  else if (k == "gp_col") g_gp_col = value;
  else if (k == "pool_proj") g_pool_proj = value;
  else if (k == "gp_col_leader") g_gp_col_leader = value;
  else {
    set_error("roma_tuning: unknown key " + k);
    return ROMA_ERR_ARG;
  }
  return 0;
}

int roma_vit_forward(const roma_vit_args_t* a, void* stream) {
  ROMA_REQUIRE(a, "roma_vit_forward: null argument");
  ROMA_REQUIRE(a->act == ROMA_F32 || a->act == ROMA_H16_CODE, "roma_vit_forward: act must be ROMA_F32 or this library's 16-bit code");
  return vit_forward(*a, S(stream));
}

int roma_op_convert_from_bf16(const void* in_bf16, void* out_h16, long n, void* stream) {
  ROMA_REQUIRE(in_bf16 && out_h16, "roma_op_convert_from_bf16: null pointer");
  return convert_from_bf16_launch(in_bf16, out_h16, n, S(stream));
}

long roma_debug_gemm_trace(unsigned int* dst_host, long nbytes) {
  ROMA_REQUIRE(dst_host && nbytes > 0, "roma_debug_gemm_trace: null destination");
  if (hipDeviceSynchronize() != hipSuccess) return ROMA_ERR_HIP;
  const int rc = gemm8p_trace_read(dst_host, nbytes);
  return rc < 0 ? rc : std::min<long>(nbytes, (long)sizeof(unsigned) * 16 * 2 * 256 * 4);
}

int roma_profile_enable(int on) {
  if (g_mixed_handles > 0)  // see roma_tuning
    if (auto f = peer_sym<int (*)(int)>("roma_profile_enable"))
      if (int rc = f(on)) return rc;
  for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.clear();
  g_prof_on = on != 0;
  return 0;
}

long roma_profile_report(char* buf, long nbytes) {
  if (hipDeviceSynchronize() != hipSuccess) return ROMA_ERR_HIP;
  struct Agg { long calls = 0; double ms = 0, work = 0; std::string unit; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    Agg& a = agg[r.name];
    a.calls++; a.ms += ms; a.work += r.work; a.unit = r.unit;
  }
  std::string js = "{";
  bool first = true;
  for (auto& kv : agg) {
    char tmp[512];
    snprintf(tmp, sizeof tmp, "%s\"%s\": {\"calls\": %ld, \"total_ms\": %.6f, \"work\": %.6e, \"unit\": \"%s\"}", first ? "" : ", ",
             kv.first.c_str(), kv.second.calls, kv.second.ms, kv.second.work, kv.second.unit.c_str());
    js += tmp;
    first = false;
  }
  if (auto f = peer_sym<long (*)(char*, long)>("roma_profile_report")) {  // the sibling library's launches (DINOv2 of a mixed handle)
    const long n = f(nullptr, 0);
    if (n > 3) {
      std::string peer((size_t)n, '\0');
      if (f(&peer[0], n) > 0) {
        peer.resize(strlen(peer.c_str()));
        if (peer.size() > 2) js += (first ? "" : ", ") + peer.substr(1, peer.size() - 2);
      }
    }
  }
  js += "}";
  if (!buf) return (long)js.size() + 1;
  if (nbytes < (long)js.size() + 1) return ROMA_ERR_ARG;
  memcpy(buf, js.c_str(), js.size() + 1);
  return (long)js.size() + 1;
}

// ------------------------------------------------------------------------------------ operators
int roma_op_local_corr(const void* feature0, const void* feature1, const float* warp, void* out, int B, int H, int W,
                       int C, int K, int nearest, int dt_in, int dt_out, void* stream) {
  LocalCorrArgs a;
  a.nearest = nearest ? 1 : 0;
  a.f0 = feature0; a.f1 = feature1; a.warp = warp; a.out = out; a.B = B; a.H = H; a.W = W; a.C = C; a.K = K;
  a.ld0 = C; a.ld1 = C; a.ldo = K; a.nimg = B; a.f1_shift = 0; a.scale = 1.f; a.in_dt = DT(dt_in); a.out_dt = DT(dt_out);
  return local_corr_general_launch(a, S(stream));
}

int roma_op_local_corr_window(const void* feature0, const void* feature1, const float* warp, void* out, int B, int H,
                              int W, int C, int radius, float scale, long ldo, int dt_in, int dt_out, void* stream) {
  LocalCorrArgs a;
  a.f0 = feature0; a.f1 = feature1; a.warp = warp; a.out = out; a.B = B; a.H = H; a.W = W; a.C = C; a.radius = radius;
  a.ld0 = C; a.ld1 = C; a.ldo = ldo; a.nimg = B; a.f1_shift = 0; a.scale = scale; a.in_dt = DT(dt_in); a.out_dt = DT(dt_out);
  ROMA_REQUIRE(ldo >= (2 * radius + 1) * (2 * radius + 1), "local_corr_window: ldo < K");
  return local_corr_window_launch(a, S(stream));
}

int roma_op_gemm(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, int batch,
                 long sA, long sW, long sC, const float* bias, const float* scale, const float* res, long ldr, int act,
                 float alpha, int dt_in, int dt_out, void* stream) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.batch = batch;
  g.sA = sA; g.sW = sW; g.sC = sC; g.bias = bias; g.scale = scale; g.res = res; g.ldr = ldr; g.sR = sC; g.act = act;
  g.alpha = alpha; g.in_dt = DT(dt_in); g.out_dt = DT(dt_out);
  return gemm_launch(g, S(stream));
}

int roma_op_conv3x3(const void* in, const void* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                    int relu, int dt, void* stream) {
  GemmArgs g;
  g.A = in; g.W = w; g.ldw = 9 * Cin; g.C = out; g.ldc = Cout; g.M = B * H * W; g.N = Cout; g.K = 9 * Cin;
  g.in_dt = DT(dt); g.out_dt = DT(dt); g.bias = bias; g.act = relu ? ACT_RELU : ACT_NONE;
  g.conv_h = H; g.conv_w = W; g.conv_c = Cin;
  return gemm_launch(g, S(stream));
}

int roma_op_conv3x3_slab(const void* in, const void* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                         int relu, int dt, void* stream) {
  GemmArgs g;
  g.A = in; g.W = w; g.ldw = 9 * Cin; g.C = out; g.ldc = Cout; g.M = B * H * W; g.N = Cout; g.K = 9 * Cin;
  g.in_dt = DT(dt); g.out_dt = DT(dt); g.bias = bias; g.act = relu ? ACT_RELU : ACT_NONE;
  g.conv_h = H; g.conv_w = W; g.conv_c = Cin; g.conv_korder = 1;
  return gemm_launch(g, S(stream));
}

int roma_op_attention(const void* q, const void* k, const void* vt, void* out, int B, int heads, int N, int npad, int hd,
                      int dt_in, int dt_out, void* stream) {
  AttnArgs a;
  a.q = q; a.k = k; a.vt = vt; a.out = out; a.B = B; a.heads = heads; a.N = N; a.npad = npad; a.hd = hd;
  a.ldo = (long)heads * hd; a.in_dt = DT(dt_in); a.out_dt = DT(dt_out);
  return attention_launch(a, S(stream));
}

int roma_op_qkv_scatter_gemm(const void* A, const void* W, const float* bias, void* q, void* k, void* vt, int B, int N,
                             int npad, int heads, int hd, int K, int dt_in, int dt_out, void* stream) {
  GemmArgs g;
  g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = B * N; g.N = 3 * heads * hd; g.K = K; g.in_dt = DT(dt_in);
  g.out_dt = DT(dt_out); g.bias = bias; g.mode = EPI_QKV; g.q = q; g.k = k; g.vt = vt; g.heads = heads; g.hd = hd;
  g.ntok = N; g.npad = npad; g.qscale = 1.0f / sqrtf((float)hd);
  return gemm_launch(g, S(stream));
}

int roma_op_layernorm(const float* x, const float* w, const float* b, void* out, long M, int D, float eps, int dt_out,
                      void* stream) {
  return layernorm_launch(x, w, b, out, M, D, eps, DT(dt_out), S(stream));
}

int roma_op_layernorm_dt(const void* x, int dt_in, const float* w, const float* b, void* out, long M, int D, float eps,
                         int dt_out, void* stream) {
  return layernorm_launch_dt(x, DT(dt_in), w, b, out, M, D, eps, DT(dt_out), S(stream));
}

int roma_op_gemm_res_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K,
                          const float* bias, const float* scale, const void* res, long ldr, void* stream) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.bias = bias; g.scale = scale; g.res_bf16 = res; g.ldr = ldr; g.in_dt = DT_BF16; g.out_dt = DT_BF16;
  return gemm_launch(g, S(stream));
}

int roma_op_cholesky_solve_t(float* A, float* Ft, float* LT, float* Linv, float* LinvT, int n, int d, int batch,
                             void* stream) {
  // Ft == A + n * n with batch > 1 cannot be the dense [batch, n, n] + [batch, d, n] layout (Ft would overlap the second
  // matrix): it is the augmented layout of the GP - one (n + d) x n matrix per item, items (n + d) * n floats apart
  if (batch > 1 && Ft == A + (long)n * n)
    return cholesky_solve_t(A, Ft, LT, Linv, LinvT, n, d, batch, S(stream), (long)(n + d) * n, (long)(n + d) * n);
  return cholesky_solve_t(A, Ft, LT, Linv, LinvT, n, d, batch, S(stream));
}

int roma_op_gp(const void* x, const void* y, const float* pos_w, const float* pos_b, float* mu, int b, int h, int w, int dt,
               void* stream) {
  ROMA_REQUIRE(x && y && pos_w && pos_b && mu && b > 0 && h > 0 && w > 0, "roma_op_gp: bad arguments");
  hipStream_t st = S(stream);
  const int n = h * w;
  const size_t esz = DT(dt) == DT_F32 ? 4 : 2;
  // one feature buffer [2b, n, 512]: images [0, b) = x (queries), [b, 2b) = y (supports); non-symmetric pairing
  Arena plan;
  plan.alloc((size_t)2 * b * n * 512 * esz);
  if (int rc = gp_posterior(nullptr, 512, DT(dt), b, false, h, w, pos_w, pos_b, mu, 512, plan, st, true)) return rc;
  Arena ar;
  ar.cap = plan.peak + 4096;
  ar.dry = false;
  ROMA_CHECK_HIP(hipMallocAsync(reinterpret_cast<void**>(&ar.base), ar.cap, st));
  char* pf = static_cast<char*>(ar.alloc((size_t)2 * b * n * 512 * esz));
  int rc = 0;
  if (hipMemcpyAsync(pf, x, (size_t)b * n * 512 * esz, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(pf + (size_t)b * n * 512 * esz, y, (size_t)b * n * 512 * esz, hipMemcpyDeviceToDevice, st) != hipSuccess) {
    set_error("roma_op_gp: device copy failed");
    rc = ROMA_ERR_HIP;
  }
  if (!rc) rc = gp_posterior(pf, 512, DT(dt), b, false, h, w, pos_w, pos_b, mu, 512, ar, st, false);
  (void)hipFreeAsync(ar.base, st);
  return rc;
}

int roma_op_cls_to_flow(const float* logits, long ld, float* flow, float* cert, long M, void* stream) {
  return cls_to_flow_launch(logits, ld, flow, cert, M, S(stream));
}

int roma_op_resize_bilinear(const float* in, float* out, int B, int Hin, int Win, int Hout, int Wout, int nc,
                            void* stream) {
  return resize_bilinear_launch(in, out, B, Hin, Win, Hout, Wout, nc, S(stream));
}

int roma_op_refiner_input(const void* feat, long ldf, const float* flow, void* d, long ldd, const float* emb_w,
                          const float* emb_b, int B, int H, int W, int C, int E, int Kcorr, int nimg, int shift,
                          float disp_scale, int dt, void* stream) {
  RefinerInputArgs a;
  a.feat = feat; a.ldf = ldf; a.flow = flow; a.d = d; a.ldd = ldd; a.emb_w = emb_w; a.emb_b = emb_b;
  a.B = B; a.H = H; a.W = W; a.C = C; a.E = E; a.Kcorr = Kcorr; a.nimg = nimg; a.shift = shift;
  a.disp_scale = disp_scale; a.dt = DT(dt);
  ROMA_REQUIRE(feat && flow && d && B > 0 && H > 0 && W > 0 && C > 0 && nimg > 0, "roma_op_refiner_input: bad arguments");
  ROMA_REQUIRE(ldd >= 2 * C + E + Kcorr && ldf >= C, "roma_op_refiner_input: row strides too small");
  return refiner_input_launch(a, S(stream));
}

int roma_op_dwconv5x5(const void* in, void* out, const float* w, const float* bias, int B, int H, int W, int Cp, int dt,
                      void* stream) {
  return dwconv5x5_launch(in, out, w, bias, B, H, W, Cp, DT(dt), S(stream));
}

int roma_op_refiner_block(const void* in, void* out, const float* dw_w, const float* dw_b, const void* pw,
                          const float* pw_b, int B, int H, int W, int Cp, int dt, void* stream) {
  if (refiner_block_wide_supported(Cp, DT(dt))) {  // C = 576: the wide fused block (refiner_block_wide.hip)
    const int rc = refiner_block_wide_try_launch(in, out, dw_w, dw_b, pw, Cp, pw_b, B, H, W, Cp, DT(dt), S(stream), true);
    if (rc <= 0) return rc;
    set_error("roma_op_refiner_block: the C = 576 kernel declined these tensors (16-byte aligned bf16 in / out / weights are required)");
    return ROMA_ERR_ARG;
  }
  return refiner_block_launch(in, out, dw_w, dw_b, pw, Cp, pw_b, B, H, W, Cp, DT(dt), S(stream));
}

int roma_op_refiner_block_final(const void* in, float* delta, const float* dw_w, const float* dw_b, const void* pw_final,
                                const float* bias_final, int B, int H, int W, int Cp, int dt, void* stream) {
  ROMA_REQUIRE(in && delta && dw_w && dw_b && pw_final && bias_final, "roma_op_refiner_block_final: null pointer");
  return refiner_block_final_launch(in, delta, dw_w, dw_b, pw_final, Cp, bias_final, B, H, W, Cp, DT(dt), S(stream));
}

int roma_op_refiner_apply_delta(const float* delta, float* flow, float* cert, long M, float sx, float sy, void* stream) {
  ROMA_REQUIRE(delta && flow && cert && M >= 0, "roma_op_refiner_apply_delta: null pointer");
  return refiner_apply_delta_launch(delta, flow, cert, M, sx, sy, S(stream));
}

int roma_op_kde(const float* x, long n, int down, float std, int half_inputs, float* density, void* stream) {
  return kde_launch(x, n, down, std, half_inputs, density, S(stream));
}

int roma_op_sample_warp_at(const float* warp, const float* cert, int H, int W, const float* xa, long n, float* xa_to_b,
                           float* cert_a, void* stream) {
  return sample_warp_at_launch(warp, cert, H, W, xa, n, xa_to_b, cert_a, S(stream));
}

int roma_op_mutual_nn(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                      int* match_b, void* ws_a, void* ws_b, void* stream) {
  return mutual_nn_launch(a, na, b, nb, cert_a, cert_th, max_dist, match_b, static_cast<unsigned long long*>(ws_a),
                          static_cast<unsigned long long*>(ws_b), S(stream));
}

int roma_op_mutual_nn_count(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                            void* ws_a, void* ws_b, long long* offsets, void* stream) {
  return mutual_nn_count_launch(a, na, b, nb, cert_a, cert_th, max_dist, static_cast<unsigned long long*>(ws_a),
                                static_cast<unsigned long long*>(ws_b), offsets, S(stream));
}
int roma_op_mutual_nn_fill(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                           const void* ws_a, const void* ws_b, long long* offsets, long long* pairs, void* stream) {
  return mutual_nn_fill_launch(a, na, b, nb, cert_a, cert_th, max_dist, static_cast<const unsigned long long*>(ws_a),
                               static_cast<const unsigned long long*>(ws_b), offsets, pairs, S(stream));
}

long roma_op_multinomial_workspace(long n, long k) { return (long)multinomial_workspace_bytes(n, k); }
int roma_op_multinomial(const float* weights, long n, long k, unsigned long long seed, long long* out_indices, void* workspace,
                        long workspace_bytes, void* stream) {
  return multinomial_launch(weights, n, k, seed, out_indices, workspace, (size_t)workspace_bytes, S(stream));
}
// ---- Tiny RoMa matcher side (tiny.hip)
int roma_op_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, void* stream) {
  return nchw_to_nhwc_launch(in, out, B, C, H, W, S(stream));
}
int roma_op_tiny_pos_embed(const float* corr_volume, float* out, int B, int H1, int W1, int H0, int W0, int exact_softmax,
                           void* stream) {
  return tiny_pos_embed_launch(corr_volume, out, B, H1, W1, H0, W0, exact_softmax, S(stream));
}
int roma_op_gray_instnorm(const float* in, float* out, int B, int H, int W, int C, float eps, void* stream) {
  return gray_instnorm_launch(in, out, B, H, W, C, eps, S(stream));
}
int roma_op_conv2d_nhwc(const float* in, const float* w, const float* bias, const float* res, float* out, int B, int H, int W,
                        int Cin, int Cout, int K, int stride, int pad, int relu, void* stream) {
  return conv2d_nhwc_launch(in, w, bias, res, out, B, H, W, Cin, Cout, K, stride, pad, relu, S(stream));
}
int roma_op_avgpool_nhwc(const float* in, float* out, int B, int H, int W, int C, int k, void* stream) {
  return avgpool_nhwc_launch(in, out, B, H, W, C, k, S(stream));
}
int roma_op_add3(const float* a, const float* b, const float* c, float* out, long n, void* stream) {
  return add3_launch(a, b, c, out, n, S(stream));
}
int roma_op_tiny_matcher_input(const float* f0, const float* f1, const float* warp, int warp_channels, float* d, int B, int H,
                               int W, int H1, int W1, int C, int Cp, void* stream) {
  return tiny_matcher_input_launch(f0, f1, warp, warp_channels, d, B, H, W, H1, W1, C, Cp, S(stream));
}
int roma_op_tiny_update(const float* base, int base_channels, const float* delta, long ldd, float sx, float sy, float* out,
                        long npix, void* stream) {
  return tiny_update_launch(base, base_channels, delta, ldd, sx, sy, out, npix, S(stream));
}
int roma_op_tiny_final(const float* matches, float* warp, float* certainty, int B, int H, int W, void* stream) {
  return tiny_final_launch(matches, warp, certainty, B, H, W, S(stream));
}
int roma_op_visualize_warp(const float* warp, const float* certainty, const float* im_a, const float* im_b, int H, int W,
                           int symmetric, int im_h, int im_w, float* out, void* stream) {
  return visualize_warp_launch(warp, certainty, im_a, im_b, H, W, symmetric, im_h, im_w, out, S(stream));
}
int roma_op_fb_consistency(const float* flow_fwd, const float* flow_bwd, int B, int H, int W, float th_n, float* out,
                           void* stream) {
  return fb_consistency_launch(flow_fwd, flow_bwd, B, H, W, th_n, out, S(stream));
}

int roma_op_maxpool2x2(const void* in, void* out, int B, int H, int W, int C, int dt, void* stream) {
  return maxpool2x2_launch(in, out, B, H, W, C, DT(dt), S(stream));
}

int roma_op_pool_proj(const void* in, void* pooled, void* pf, const void* pw, long ldw, const float* pb, int N, int ldf, int B, int H,
                      int W, int C, int dt, void* stream) {
  return pool_proj_launch(in, pooled, pf, pw, ldw, pb, N, ldf, B, H, W, C, DT(dt), S(stream));
}

int roma_op_refiner_out(const void* d, long ldd, int dt, const float* w, const float* b, float* flow, float* cert, long M,
                        int Cp, float sx, float sy, void* stream) {
  ROMA_REQUIRE(d && w && b && flow && cert && M > 0 && Cp > 0 && ldd >= Cp, "roma_op_refiner_out: bad arguments");
  return refiner_out_launch(d, ldd, DT(dt), w, b, flow, cert, M, Cp, sx, sy, S(stream));
}

int roma_op_conv3x3_c3_bf16(const float* img, const void* w, const float* bias, void* out, int B, int H, int W, void* stream) {
  return conv3x3_c3_bf16_launch(img, w, bias, out, B, H, W, S(stream));
}

int roma_op_conv3x3_c3(const float* img, const float* w, const float* bias, void* out, int B, int H, int W, int dt_out,
                       void* stream) {
  return conv3x3_c3_launch(img, w, bias, out, B, H, W, DT(dt_out), S(stream));
}

}  // extern "C"
