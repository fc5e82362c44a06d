// Model object: weight intake / packing and the match() kernel schedule (see model.h).
#include "model.h"

#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <set>

#include "attention.h"
#include "conv64.h"
#include "pool_proj.h"
#include "elementwise.h"
#include "gemm.h"
#include "local_corr.h"
#include "refiner_block.h"
#include "vit.h"

namespace roma {

std::mutex g_peer_mutex;
extern void* g_peer_lib;  // api.hip: roma_tuning / roma_profile_* forward to the sibling library while a mixed handle lives
extern int g_mixed_handles;

static const int VGG_IDX[12] = {0, 3, 7, 10, 14, 17, 20, 23, 27, 30, 33, 36};
static const int VGG_CH[12] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512};
static const char* SCALES[5] = {"16", "8", "4", "2", "1"};
static const int SCALE_INT[5] = {16, 8, 4, 2, 1};
static const int PROJ_CIN[5] = {1024, 512, 256, 128, 64};
static const int PROJ_COUT[5] = {512, 512, 256, 64, 9};
static const int REF_EMB[5] = {128, 64, 32, 16, 6};
static const int REF_RAD[5] = {7, 3, 2, 0, 0};

Model::~Model() {
  for (int i = 0; i < MAX_STREAMS - 1; ++i) {
    if (side[i]) (void)hipStreamDestroy(side[i]);
    if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
  }
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  for (void* p : owned) (void)hipFree(p);
  for (auto& kv : dbg) (void)hipFree(kv.second.first);
  for (auto& kv : inject) (void)hipFree(kv.second.first);
  if (peer_lib) {
    std::lock_guard<std::mutex> lk(g_peer_mutex);
    --g_mixed_handles;
  }
}


// ROMA_MIXED: the bfloat16 build of this library (libroma_hip.so, next to this shared object) runs DINOv2.  Loaded once per
// process with RTLD_LOCAL: both libraries export the same C ABI, each keeps its own symbols.
int Model::load_peer() {
  static void* lib = nullptr;
  static int (*fwd)(const roma_vit_args_t*, void*) = nullptr;
  std::lock_guard<std::mutex> lk(g_peer_mutex);  // handles may be created from several threads
  if (!lib) {
    ROMA_REQUIRE(roma_h16_format() == ROMA_F16, "ROMA_MIXED is a mode of the binary16 build (libroma_hip_f16.so)");
    Dl_info info;
    ROMA_REQUIRE(dladdr(reinterpret_cast<void*>(&roma_vit_forward), &info) && info.dli_fname, "ROMA_MIXED: cannot locate this library");
    std::string path(info.dli_fname);
    const size_t slash = path.find_last_of('/');
    path = (slash == std::string::npos ? std::string("") : path.substr(0, slash + 1)) + "libroma_hip.so";
    // RTLD_LOCAL: its C ABI must not shadow ours; RTLD_DEEPBIND on top of the link-time -Bsymbolic (csrc/Makefile): the
    // sibling's own definitions win over anything in the global scope - this library, when a C program linked it directly
    void* l = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (!l) {
      set_error("ROMA_MIXED: cannot load the bfloat16 library " + path + ": " + (dlerror() ? dlerror() : "?"));
      return ROMA_ERR_STATE;
    }
    auto fmt = reinterpret_cast<int (*)(void)>(dlsym(l, "roma_h16_format"));
    auto self = reinterpret_cast<int (*)(void)>(dlsym(l, "roma_self_check"));
    auto abi = reinterpret_cast<int (*)(void)>(dlsym(l, "roma_abi_stamp"));
    auto f = reinterpret_cast<int (*)(const roma_vit_args_t*, void*)>(dlsym(l, "roma_vit_forward"));
    std::string why;
    if (!fmt || !f || !self || !abi) why = "it does not export roma_h16_format / roma_self_check / roma_abi_stamp / roma_vit_forward (an older build)";
    else if (fmt() != ROMA_BF16) why = "it is not the bfloat16 build";
    else if (abi() != roma_abi_stamp()) why = "its ABI stamp " + std::to_string(abi()) + " differs from this library's " + std::to_string(roma_abi_stamp()) + " (stale build: roma_vit_args_t would be misread)";
    else if (self() != ROMA_BF16) why = "its internal calls resolve into another library (symbol interposition: link both builds with -Wl,-Bsymbolic)";
    if (!why.empty()) {
      dlclose(l);
      set_error("ROMA_MIXED: " + path + " cannot serve as the bfloat16 sibling: " + why);
      return ROMA_ERR_STATE;
    }
    lib = l;
    fwd = f;
    g_peer_lib = l;
  }
  if (!peer_lib) ++g_mixed_handles;
  peer_lib = lib;
  peer_vit_forward = fwd;
  return 0;
}

int Model::debug_inject(const char* name, const void* host, size_t bytes) {
  ROMA_REQUIRE(name, "roma_debug_inject: null name");
  const std::string k(name);
  ROMA_REQUIRE(k == "gm_flow16" || k == "gm_cert16", "roma_debug_inject: unknown stage (gm_flow16, gm_cert16)");
  ROMA_CHECK_HIP(hipSetDevice(cfg.device));
  auto it = inject.find(k);
  if (it != inject.end()) {
    ROMA_CHECK_HIP(hipDeviceSynchronize());
    (void)hipFree(it->second.first);
    inject.erase(it);
  }
  if (!host || bytes == 0) return 0;
  void* p = nullptr;
  ROMA_CHECK_HIP(hipMalloc(&p, bytes));
  ROMA_CHECK_HIP(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
  inject[k] = {p, bytes};
  return 0;
}

int Model::set_tensor(const char* name, int ndim, const int64_t* shape, const void* data, int is_int64) {
  ROMA_REQUIRE(!finalized, "roma_set_tensor: model already finalized");
  ROMA_REQUIRE(name && data && ndim >= 0 && ndim <= 8, "roma_set_tensor: bad arguments");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  const long n = t.numel();
  t.data.resize((size_t)n);
  if (is_int64) {
    const int64_t* s = static_cast<const int64_t*>(data);
    for (long i = 0; i < n; ++i) t.data[(size_t)i] = (float)s[i];
  } else {
    memcpy(t.data.data(), data, (size_t)n * sizeof(float));
  }
  host[name] = std::move(t);
  return 0;
}

// ---------------------------------------------------------------- strict state-dict contract
static void expect(std::vector<std::pair<std::string, std::vector<int64_t>>>& v, const std::string& k,
                   std::vector<int64_t> s) {
  v.emplace_back(k, std::move(s));
}
static void expect_bn(std::vector<std::pair<std::string, std::vector<int64_t>>>& v, const std::string& p, int c) {
  expect(v, p + ".weight", {c});
  expect(v, p + ".bias", {c});
  expect(v, p + ".running_mean", {c});
  expect(v, p + ".running_var", {c});
  expect(v, p + ".num_batches_tracked", {});
}

int Model::check_contract() {
  std::vector<std::pair<std::string, std::vector<int64_t>>> e;
  int cin = 3;
  for (int i = 0; i < 12; ++i) {
    const std::string p = "encoder.cnn.layers." + std::to_string(VGG_IDX[i]);
    expect(e, p + ".weight", {VGG_CH[i], cin, 3, 3});
    expect(e, p + ".bias", {VGG_CH[i]});
    expect_bn(e, "encoder.cnn.layers." + std::to_string(VGG_IDX[i] + 1), VGG_CH[i]);
    cin = VGG_CH[i];
  }
  auto vit = [&](const std::string& p, bool qkv_bias, bool ls) {
    expect(e, p + ".norm1.weight", {1024});
    expect(e, p + ".norm1.bias", {1024});
    expect(e, p + ".attn.qkv.weight", {3072, 1024});
    if (qkv_bias) expect(e, p + ".attn.qkv.bias", {3072});
    expect(e, p + ".attn.proj.weight", {1024, 1024});
    expect(e, p + ".attn.proj.bias", {1024});
    if (ls) expect(e, p + ".ls1.gamma", {1024});
    expect(e, p + ".norm2.weight", {1024});
    expect(e, p + ".norm2.bias", {1024});
    expect(e, p + ".mlp.fc1.weight", {4096, 1024});
    expect(e, p + ".mlp.fc1.bias", {4096});
    expect(e, p + ".mlp.fc2.weight", {1024, 4096});
    expect(e, p + ".mlp.fc2.bias", {1024});
    if (ls) expect(e, p + ".ls2.gamma", {1024});
  };
  for (int i = 0; i < 5; ++i) vit("decoder.embedding_decoder.blocks." + std::to_string(i), false, false);
  expect(e, "decoder.embedding_decoder.to_out.weight", {4097, 1024});
  expect(e, "decoder.embedding_decoder.to_out.bias", {4097});
  expect(e, "decoder.gps.16.pos_conv.weight", {512, 2, 1, 1});
  expect(e, "decoder.gps.16.pos_conv.bias", {512});
  for (int s = 0; s < 5; ++s) {
    const std::string p = std::string("decoder.proj.") + SCALES[s];
    expect(e, p + ".0.weight", {PROJ_COUT[s], PROJ_CIN[s], 1, 1});
    expect(e, p + ".0.bias", {PROJ_COUT[s]});
    expect_bn(e, p + ".1", PROJ_COUT[s]);
  }
  for (int s = 0; s < 5; ++s) {
    const int K = REF_RAD[s] ? (2 * REF_RAD[s] + 1) * (2 * REF_RAD[s] + 1) : 0;
    const int C = 2 * PROJ_COUT[s] + REF_EMB[s] + K;
    const std::string p = std::string("decoder.conv_refiner.") + SCALES[s];
    for (int b = 0; b < 9; ++b) {
      const std::string bp = b == 0 ? p + ".block1" : p + ".hidden_blocks." + std::to_string(b - 1);
      expect(e, bp + ".0.weight", {C, 1, 5, 5});
      expect(e, bp + ".0.bias", {C});
      expect_bn(e, bp + ".1", C);
      expect(e, bp + ".3.weight", {C, C, 1, 1});
      expect(e, bp + ".3.bias", {C});
    }
    expect(e, p + ".out_conv.weight", {3, C, 1, 1});
    expect(e, p + ".out_conv.bias", {3});
    expect(e, p + ".disp_emb.weight", {REF_EMB[s], 2, 1, 1});
    expect(e, p + ".disp_emb.bias", {REF_EMB[s]});
  }
  // DINOv2 dict under the "dinov2." prefix
  expect(e, "dinov2.cls_token", {1, 1, 1024});
  expect(e, "dinov2.pos_embed", {1, 1370, 1024});
  expect(e, "dinov2.mask_token", {1, 1024});
  expect(e, "dinov2.patch_embed.proj.weight", {1024, 3, 14, 14});
  expect(e, "dinov2.patch_embed.proj.bias", {1024});
  for (int i = 0; i < 24; ++i) vit("dinov2.blocks." + std::to_string(i), true, true);
  expect(e, "dinov2.norm.weight", {1024});
  expect(e, "dinov2.norm.bias", {1024});

  std::set<std::string> seen;
  for (auto& kv : e) {
    auto it = host.find(kv.first);
    if (it == host.end()) {
      set_error("roma_finalize: missing key in state_dict: " + kv.first);
      return ROMA_ERR_STATE;
    }
    if (it->second.shape != kv.second) {
      set_error("roma_finalize: size mismatch for " + kv.first);
      return ROMA_ERR_STATE;
    }
    seen.insert(kv.first);
  }
  for (auto& kv : host)
    if (!seen.count(kv.first)) {
      set_error("roma_finalize: unexpected key in state_dict: " + kv.first);
      return ROMA_ERR_STATE;
    }
  return 0;
}

// ---------------------------------------------------------------- uploads
template <typename F>
int Model::upload_f32(const std::vector<float>& v, F** out) {
  void* p = nullptr;
  ROMA_CHECK_HIP(hipMalloc(&p, std::max<size_t>(v.size(), 4) * sizeof(float)));
  owned.push_back(p);
  ROMA_CHECK_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = static_cast<F*>(p);
  return 0;
}

int Model::upload_act(const std::vector<float>& v, void** out) {
  if (act_dt == DT_F32) {
    float* p = nullptr;
    if (int rc = upload_f32(v, &p)) return rc;
    *out = p;
    return 0;
  }
  std::vector<bf16_t> h(v.size());
  if (pack_as_bf16) {
    for (size_t i = 0; i < v.size(); ++i) h[i] = f32_to_bfloat16_bits(v[i]);  // ROMA_MIXED: the DINOv2 weights of the bf16 library
  } else {
    for (size_t i = 0; i < v.size(); ++i) h[i] = f32_to_bf16(v[i]);
  }
  void* p = nullptr;
  ROMA_CHECK_HIP(hipMalloc(&p, std::max<size_t>(h.size(), 8) * sizeof(bf16_t)));
  owned.push_back(p);
  ROMA_CHECK_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
  *out = p;
  return 0;
}

// w: [N][K] row-major; pads K to a multiple of 8
int Model::make_lin(const std::vector<float>& w, const std::vector<float>* b, int N, int K, Lin* out) {
  const int ldw = (int)round_up(K, 8);
  std::vector<float> wp((size_t)N * ldw, 0.f);
  for (int n = 0; n < N; ++n) memcpy(&wp[(size_t)n * ldw], &w[(size_t)n * K], (size_t)K * sizeof(float));
  out->N = N;
  out->K = ldw;
  out->ldw = ldw;
  if (int rc = upload_act(wp, &out->w)) return rc;
  out->b = nullptr;
  if (b)
    if (int rc = upload_f32(*b, &out->b)) return rc;
  return 0;
}

static void bn_fold(const std::map<std::string, HostTensor>& h, const std::string& p, std::vector<float>& s,
                    std::vector<float>& t) {
  const auto& g = h.at(p + ".weight").data;
  const auto& be = h.at(p + ".bias").data;
  const auto& mu = h.at(p + ".running_mean").data;
  const auto& var = h.at(p + ".running_var").data;
  s.resize(g.size());
  t.resize(g.size());
  for (size_t i = 0; i < g.size(); ++i) {
    s[i] = g[i] / sqrtf(var[i] + 1e-5f);
    t[i] = be[i] - mu[i] * s[i];
  }
}

static inline float cubic1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
static inline float cubic2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// DINOv2 interpolate_pos_encoding (romatch/models/transformer/dinov2.py:166-190): bicubic, A=-0.75, with the
// scale_factor=(h0+0.1)/37 quirk (coordinates map with 37/(h0+0.1), not 37/h0).
static std::vector<float> resize_pos_embed(const std::vector<float>& pos, int th, int tw) {
  const int M = 37, D = 1024;
  std::vector<float> out((size_t)(1 + th * tw) * D);
  memcpy(out.data(), pos.data(), D * sizeof(float));
  if (th == M && tw == M) {
    memcpy(out.data(), pos.data(), out.size() * sizeof(float));
    return out;
  }
  const float sh = (float)(1.0 / ((th + 0.1) / 37.0)), sw = (float)(1.0 / ((tw + 0.1) / 37.0));
  std::vector<int> iy(th * 4), ix(tw * 4);
  std::vector<float> wy(th * 4), wx(tw * 4);
  auto prep = [&](int n, float scale, std::vector<int>& idx, std::vector<float>& wgt) {
    for (int o = 0; o < n; ++o) {
      const float src = scale * ((float)o + 0.5f) - 0.5f;
      const float fl = floorf(src);
      float t = src - fl;
      t = std::min(std::max(t, 0.f), 1.f);
      const int i0 = (int)fl;
      const float c[4] = {cubic2(t + 1.f), cubic1(t), cubic1(1.f - t), cubic2(2.f - t)};
      for (int j = 0; j < 4; ++j) {
        idx[o * 4 + j] = std::max(std::min(i0 - 1 + j, M - 1), 0);
        wgt[o * 4 + j] = c[j];
      }
    }
  };
  prep(th, sh, iy, wy);
  prep(tw, sw, ix, wx);
  const float* pp = pos.data() + D;  // patch part [37*37][D]
  for (int y = 0; y < th; ++y)
    for (int x = 0; x < tw; ++x) {
      float* o = &out[(size_t)(1 + y * tw + x) * D];
      for (int d = 0; d < D; ++d) o[d] = 0.f;
      for (int a = 0; a < 4; ++a) {
        for (int b = 0; b < 4; ++b) {
          const float w = wy[y * 4 + a] * wx[x * 4 + b];
          const float* src = pp + (size_t)(iy[y * 4 + a] * M + ix[x * 4 + b]) * D;
          for (int d = 0; d < D; ++d) o[d] += w * src[d];
        }
      }
    }
  return out;
}

int Model::pack_weights() {
  auto H = [&](const std::string& k) -> const std::vector<float>& { return host.at(k).data; };
  // ---- VGG19-BN: fold BN, repack to [cout][(ky*3+kx)*cin + ci]
  int cin = 3;
  for (int i = 0; i < 12; ++i) {
    const int cout = VGG_CH[i];
    const std::string p = "encoder.cnn.layers." + std::to_string(VGG_IDX[i]);
    std::vector<float> s, t;
    bn_fold(host, "encoder.cnn.layers." + std::to_string(VGG_IDX[i] + 1), s, t);
    const auto& w = H(p + ".weight");
    const auto& b = H(p + ".bias");
    std::vector<float> bf(cout);
    for (int o = 0; o < cout; ++o) bf[o] = b[o] * s[o] + t[o];
    if (i == 0) {
      std::vector<float> wp(27 * 64);
      for (int o = 0; o < 64; ++o)
        for (int ci = 0; ci < 3; ++ci)
          for (int k = 0; k < 9; ++k) wp[(ci * 9 + k) * 64 + o] = w[(o * 3 + ci) * 9 + k] * s[o];
      if (int rc = upload_f32(wp, &c1_w)) return rc;
      if (int rc = upload_f32(bf, &c1_b)) return rc;
      // MFMA form of the first layer: [64][32] with k = ci*9 + ky*3 + kx (27 real taps, zero padded to 32)
      std::vector<float> wg(64 * 27);
      for (int o = 0; o < 64; ++o)
        for (int k = 0; k < 27; ++k) wg[o * 27 + k] = w[o * 27 + k] * s[o];
      if (int rc = make_lin(wg, &bf, 64, 27, &vgg[0])) return rc;
    } else {
      std::vector<float> wp((size_t)cout * 9 * cin);
      vgg_korder[i] = (vgg_slab_major && act_dt != DT_F32 && cout >= 256 && cin % 64 == 0) ? 1 : 0;
      for (int o = 0; o < cout; ++o)
        for (int ci = 0; ci < cin; ++ci)
          for (int k = 0; k < 9; ++k) {
            const size_t kk = vgg_korder[i] ? ((size_t)(ci / 64) * 9 + k) * 64 + ci % 64 : (size_t)k * cin + ci;
            wp[(size_t)o * 9 * cin + kk] = w[((size_t)o * cin + ci) * 9 + k] * s[o];
          }
      if (int rc = make_lin(wp, &bf, cout, 9 * cin, &vgg[i])) return rc;
    }
    vgg_cin[i] = cin;
    vgg_cout[i] = cout;
    cin = cout;
  }
  // ---- ViT blocks
  auto pack_vit = [&](const std::string& p, bool qkv_bias, bool ls, VitBlockW& o) -> int {
    if (int rc = upload_f32(H(p + ".norm1.weight"), &o.ln1w)) return rc;
    if (int rc = upload_f32(H(p + ".norm1.bias"), &o.ln1b)) return rc;
    if (int rc = upload_f32(H(p + ".norm2.weight"), &o.ln2w)) return rc;
    if (int rc = upload_f32(H(p + ".norm2.bias"), &o.ln2b)) return rc;
    if (int rc = make_lin(H(p + ".attn.qkv.weight"), qkv_bias ? &H(p + ".attn.qkv.bias") : nullptr, 3072, 1024, &o.qkv)) return rc;
    if (int rc = make_lin(H(p + ".mlp.fc1.weight"), &H(p + ".mlp.fc1.bias"), 4096, 1024, &o.fc1)) return rc;
    if (ls && act_dt == DT_BF16) {
      // LayerScale (layers/layer_scale.py:27-28) folded into the branch's last linear: x += g * (W a + b) = (g W) a + g b.
      // In the throughput mode the product g W is rounded to bf16 once at pack time instead of the branch output being
      // scaled in f32 before its bf16 rounding - the same 2^-9 class of rounding - and the GEMM epilogue needs no
      // per-column scale vector (32 registers it does not have next to 128 accumulators).  The exact-f32 mode keeps the
      // reference's order of operations.
      auto fold = [&](const std::vector<float>& w, const std::vector<float>& b, const std::vector<float>& g, int N, int K,
                      Lin* out) -> int {
        std::vector<float> wf(w), bf(b);
        for (int n = 0; n < N; ++n) {
          for (int k = 0; k < K; ++k) wf[(size_t)n * K + k] *= g[n];
          bf[n] *= g[n];
        }
        return make_lin(wf, &bf, N, K, out);
      };
      if (int rc = fold(H(p + ".attn.proj.weight"), H(p + ".attn.proj.bias"), H(p + ".ls1.gamma"), 1024, 1024, &o.proj)) return rc;
      if (int rc = fold(H(p + ".mlp.fc2.weight"), H(p + ".mlp.fc2.bias"), H(p + ".ls2.gamma"), 1024, 4096, &o.fc2)) return rc;
      return 0;
    }
    if (ls) {
      if (int rc = upload_f32(H(p + ".ls1.gamma"), &o.ls1)) return rc;
      if (int rc = upload_f32(H(p + ".ls2.gamma"), &o.ls2)) return rc;
    }
    if (int rc = make_lin(H(p + ".attn.proj.weight"), &H(p + ".attn.proj.bias"), 1024, 1024, &o.proj)) return rc;
    if (int rc = make_lin(H(p + ".mlp.fc2.weight"), &H(p + ".mlp.fc2.bias"), 1024, 4096, &o.fc2)) return rc;
    return 0;
  };
  pack_as_bf16 = mixed;  // ROMA_MIXED: DINOv2's GEMM operands are bfloat16 (it runs in the bf16 library)
  for (int i = 0; i < 24; ++i)
    if (int rc = pack_vit("dinov2.blocks." + std::to_string(i), true, true, dino[i])) return rc;
  if (int rc = make_lin(H("dinov2.patch_embed.proj.weight"), &H("dinov2.patch_embed.proj.bias"), 1024, 588, &patch)) return rc;
  pack_as_bf16 = false;
  for (int i = 0; i < 5; ++i)
    if (int rc = pack_vit("decoder.embedding_decoder.blocks." + std::to_string(i), false, false, tdec[i])) return rc;
  {
    std::vector<float> cls(H("dinov2.cls_token"));
    if (int rc = upload_f32(cls, &cls_tok)) return rc;
    std::vector<float> pe = resize_pos_embed(H("dinov2.pos_embed"), cfg.coarse_h / 14, cfg.coarse_w / 14);
    if (int rc = upload_f32(pe, &pos_emb)) return rc;
  }
  if (int rc = upload_f32(H("dinov2.norm.weight"), &dino_nw)) return rc;
  if (int rc = upload_f32(H("dinov2.norm.bias"), &dino_nb)) return rc;
  {
    // 4097 = 64 * 64 anchors + 1 is not a multiple of 4, which keeps the f32 row writers of the 8-phase kernels away (the launch
    // fell to the classic 128 x 128 kernel: 4 launches of 183 us per step at 0.4 PFLOP/s).  Three zero rows (weights and bias)
    // make it 4100: logits columns 4097 .. 4099 come out 0 and nobody reads them (ldl = 4104 below; cls_to_flow takes 4097).
    const auto& w0 = H("decoder.embedding_decoder.to_out.weight");
    const auto& b0 = H("decoder.embedding_decoder.to_out.bias");
    std::vector<float> wpad((size_t)4100 * 1024, 0.f), bpad(4100, 0.f);
    memcpy(wpad.data(), w0.data(), (size_t)4097 * 1024 * sizeof(float));
    memcpy(bpad.data(), b0.data(), (size_t)4097 * sizeof(float));
    if (int rc = make_lin(wpad, &bpad, 4100, 1024, &to_out)) return rc;
  }
  if (int rc = upload_f32(H("decoder.gps.16.pos_conv.weight"), &gp_w)) return rc;
  if (int rc = upload_f32(H("decoder.gps.16.pos_conv.bias"), &gp_b)) return rc;
  // ---- proj heads (conv1x1 + BN folded)
  for (int s = 0; s < 5; ++s) {
    const std::string p = std::string("decoder.proj.") + SCALES[s];
    std::vector<float> sc, sh;
    bn_fold(host, p + ".1", sc, sh);
    const int co = PROJ_COUT[s], ci = PROJ_CIN[s];
    std::vector<float> w(H(p + ".0.weight")), b(co);
    for (int o = 0; o < co; ++o) {
      for (int c = 0; c < ci; ++c) w[(size_t)o * ci + c] *= sc[o];
      b[o] = H(p + ".0.bias")[o] * sc[o] + sh[o];
    }
    if (int rc = make_lin(w, &b, co, ci, &proj[s])) return rc;
  }
  // ---- ConvRefiners
  for (int s = 0; s < 5; ++s) {
    RefinerW& r = ref[s];
    r.Cf = PROJ_COUT[s];
    r.E = REF_EMB[s];
    r.radius = REF_RAD[s];
    r.K = r.radius ? (2 * r.radius + 1) * (2 * r.radius + 1) : 0;
    r.C = 2 * r.Cf + r.E + r.K;
    // channel padding: whole 128-byte rows for the wide (MFMA-bound) refiners so every GEMM slab row is one aligned
    // cache line; 16-byte granularity for the narrow HBM-bound ones (144, 24 channels)
    r.Cp = (int)(r.C >= 256 ? round_up(r.C, 64) : round_up(r.C, 8));
    const std::string p = std::string("decoder.conv_refiner.") + SCALES[s];
    if (int rc = upload_f32(H(p + ".disp_emb.weight"), &r.emb_w)) return rc;
    if (int rc = upload_f32(H(p + ".disp_emb.bias"), &r.emb_b)) return rc;
    for (int b = 0; b < 9; ++b) {
      const std::string bp = b == 0 ? p + ".block1" : p + ".hidden_blocks." + std::to_string(b - 1);
      std::vector<float> sc, sh;
      bn_fold(host, bp + ".1", sc, sh);
      const auto& w = H(bp + ".0.weight");
      const auto& bb = H(bp + ".0.bias");
      std::vector<float> dw((size_t)25 * r.Cp, 0.f), db((size_t)r.Cp, 0.f);
      for (int c = 0; c < r.C; ++c) {
        for (int t = 0; t < 25; ++t) dw[(size_t)t * r.Cp + c] = w[(size_t)c * 25 + t] * sc[c];
        db[c] = bb[c] * sc[c] + sh[c];
      }
      if (int rc = upload_f32(dw, &r.dw_w[b])) return rc;
      if (int rc = upload_f32(db, &r.dw_b[b])) return rc;
      const auto& pw = H(bp + ".3.weight");
      std::vector<float> pwp((size_t)r.Cp * r.Cp, 0.f), pb((size_t)r.Cp, 0.f);
      for (int o = 0; o < r.C; ++o) {
        memcpy(&pwp[(size_t)o * r.Cp], &pw[(size_t)o * r.C], (size_t)r.C * sizeof(float));
        pb[o] = H(bp + ".3.bias")[o];
      }
      if (int rc = make_lin(pwp, &pb, r.Cp, r.Cp, &r.pw[b])) return rc;
    }
    std::vector<float> ow((size_t)3 * r.Cp, 0.f);
    for (int o = 0; o < 3; ++o)
      memcpy(&ow[(size_t)o * r.Cp], &H(p + ".out_conv.weight")[(size_t)o * r.C], (size_t)r.C * sizeof(float));
    if (int rc = upload_f32(ow, &r.out_w)) return rc;
    if (int rc = upload_f32(H(p + ".out_conv.bias"), &r.out_b)) return rc;
    {  // out_conv o (last block's 1x1): d -> out_w (pw8 d + b8) + out_b = (out_w pw8) d + (out_w b8 + out_b), accumulated in f64.
      // pw8 enters as the kernels would see it (rounded to the 16-bit storage format in those modes, like autocast's weight cast)
      const std::string bp = p + ".hidden_blocks.7";
      const auto& pw8 = H(bp + ".3.weight");
      const auto& pb8 = H(bp + ".3.bias");
      const auto& ob = H(p + ".out_conv.bias");
      std::vector<float> cw((size_t)3 * r.Cp, 0.f), cb(4, 0.f);
      for (int o = 0; o < 3; ++o) {
        for (int k = 0; k < r.C; ++k) {
          double acc = 0.0;
          for (int n = 0; n < r.C; ++n) {
            float wv = pw8[(size_t)n * r.C + k];
            if (act_dt != DT_F32) wv = bf16_to_f32(f32_to_bf16(wv));
            acc += (double)ow[(size_t)o * r.Cp + n] * (double)wv;
          }
          cw[(size_t)o * r.Cp + k] = (float)acc;
        }
        double accb = (double)ob[o];
        for (int n = 0; n < r.C; ++n) accb += (double)ow[(size_t)o * r.Cp + n] * (double)pb8[n];
        cb[o] = (float)accb;
      }
      if (int rc = upload_f32(cw, &r.oc_w)) return rc;
      if (int rc = upload_f32(cb, &r.oc_b)) return rc;
      if (act_dt != DT_F32) {  // head + remainder split for the MFMA of the FINAL fused blocks
        std::vector<float> hl((size_t)8 * r.Cp, 0.f), fb((size_t)r.Cp, 0.f);
        for (int o = 0; o < 3; ++o) {
          for (int k = 0; k < r.Cp; ++k) {
            const float wv = cw[(size_t)o * r.Cp + k];
            const float hi = bf16_to_f32(f32_to_bf16(wv));
            hl[(size_t)o * r.Cp + k] = hi;
            hl[(size_t)(4 + o) * r.Cp + k] = wv - hi;  // (rounded to 16 bits by upload_act)
          }
          fb[o] = cb[o];
        }
        if (int rc = upload_act(hl, &r.ocf_w)) return rc;
        if (int rc = upload_f32(fb, &r.ocf_b)) return rc;
      }
    }
  }
  return 0;
}

int Model::finalize() {
  ROMA_REQUIRE(!finalized, "roma_finalize: already finalized");
  ROMA_CHECK_HIP(hipSetDevice(cfg.device));
  if (int rc = check_contract()) return rc;
  act_dt = cfg.precision == ROMA_F32 ? DT_F32 : DT_BF16;  // roma_create admitted only this build's 16-bit code (+ ROMA_MIXED)
  mixed = cfg.precision == ROMA_MIXED;
  if (mixed)
    if (int rc = load_peer()) return rc;
  if (int rc = pack_weights()) return rc;
  host.clear();
  // plan the workspace with a dry run at the largest configuration
  arena.dry = persist.dry = true;
  arena.peak = persist.peak = 0;
  const int keep_sym = cfg.symmetric, keep_up = cfg.upsample_preds;
  cfg.symmetric = 1;
  cfg.upsample_preds = cfg.upsample_h > 0 ? 1 : 0;
  int rc = match_impl(cfg.max_batch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, arena, persist);
  // side streams (model.h): every side sub-batch is at most floor(max_batch / 2) pairs; only the sizes are planned here
  Arena plan, plan_p;
  if (!rc && cfg.max_batch >= 2)
    rc = match_impl(cfg.max_batch / 2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, plan, plan_p);
  cfg.symmetric = keep_sym;
  cfg.upsample_preds = keep_up;
  if (rc) return rc;
  side_arena_bytes = plan.peak + 4096;
  side_persist_bytes = plan_p.peak + 4096;
  arena.cap = arena.peak + 4096;
  persist.cap = persist.peak + 4096;
  ROMA_CHECK_HIP(hipMalloc((void**)&arena.base, arena.cap));
  owned.push_back(arena.base);
  ROMA_CHECK_HIP(hipMalloc((void**)&persist.base, persist.cap));
  owned.push_back(persist.base);
  ROMA_CHECK_HIP(hipMemset(persist.base, 0, persist.cap));
  ROMA_CHECK_HIP(hipMemset(arena.base, 0, arena.cap));
  arena.dry = persist.dry = false;
  finalized = true;
  return 0;
}

// Side arenas, streams and events for `n` sub-batch streams, created on first use (a few GB of hipMalloc: first call only).
int Model::ensure_side_streams(int n) {
  for (int i = streams_ready - 1; i + 1 < n; ++i) {
    Arena &a = side_arena[i], &p = side_persist[i];
    a.cap = side_arena_bytes;
    p.cap = side_persist_bytes;
    ROMA_CHECK_HIP(hipMalloc((void**)&a.base, a.cap));
    owned.push_back(a.base);
    ROMA_CHECK_HIP(hipMalloc((void**)&p.base, p.cap));
    owned.push_back(p.base);
    ROMA_CHECK_HIP(hipMemset(p.base, 0, p.cap));
    ROMA_CHECK_HIP(hipMemset(a.base, 0, a.cap));
    ROMA_CHECK_HIP(hipDeviceSynchronize());
    a.dry = p.dry = false;
    ROMA_CHECK_HIP(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
    ROMA_CHECK_HIP(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
    if (!ev_fork) ROMA_CHECK_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    streams_ready = i + 2;
  }
  return 0;
}

// diagnostics (tools/repro_mixed.py): ROMA_DEBUG_DUAL_SLOT = k keeps the sub-batch stream split on in debug mode and lets only
// sub-batch k capture its stages (the capture table is per handle, not per stream)
static const int g_dbg_dual_slot = getenv("ROMA_DEBUG_DUAL_SLOT") ? atoi(getenv("ROMA_DEBUG_DUAL_SLOT")) : -1;

int Model::dbg_save(const char* name, const void* p, size_t bytes, hipStream_t st) {
  if (!debug) return 0;
  if (g_dbg_dual_slot >= 0) {  // only the stages named in ROMA_DEBUG_ONLY (comma separated), only for the chosen sub-batch
    static const std::string only = std::string(",") + (getenv("ROMA_DEBUG_ONLY") ? getenv("ROMA_DEBUG_ONLY") : "") + ",";
    if (dbg_cur_slot != g_dbg_dual_slot || only.find(std::string(",") + name + ",") == std::string::npos) return 0;
  }
  auto it = dbg.find(name);
  if (it == dbg.end() || it->second.second != bytes) {
    if (it != dbg.end()) (void)hipFree(it->second.first);
    void* q = nullptr;
    ROMA_CHECK_HIP(hipMalloc(&q, bytes));
    dbg[name] = {q, bytes};
    it = dbg.find(name);
  }
  ROMA_CHECK_HIP(hipMemcpyAsync(it->second.first, p, bytes, hipMemcpyDeviceToDevice, st));
  return 0;
}

int Model::match(int B, const float* ima, const float* imb, const float* ima_hr, const float* imb_hr, float* warp,
                 float* cert, hipStream_t st) {
  ROMA_REQUIRE(finalized, "roma_match: call roma_finalize first");
  ROMA_REQUIRE(B >= 1 && B <= cfg.max_batch, "roma_match: batch size out of range (max_batch)");
  ROMA_REQUIRE(ima && imb && warp && cert, "roma_match: null image / output pointer");
  if (cfg.upsample_preds) {
    ROMA_REQUIRE(cfg.upsample_h > 0, "roma_match: upsample_preds set but the handle has no upsample resolution");
    ROMA_REQUIRE(ima_hr && imb_hr, "roma_match: upsample_preds requires im_A_high_res and im_B_high_res");
  }
  ROMA_CHECK_HIP(hipSetDevice(cfg.device));
  return match_streams(B, ima, imb, ima_hr, imb_hr, warp, cert, st);
}

int Model::forward(int B, const float* ima, const float* imb, const roma_forward_args_t* fw, hipStream_t st) {
  ROMA_REQUIRE(finalized, "roma_forward: call roma_finalize first");
  ROMA_REQUIRE(B >= 1 && B <= cfg.max_batch, "roma_forward: batch size out of range (max_batch)");
  ROMA_REQUIRE(ima && imb && fw, "roma_forward: null image / argument pointer");
  if (fw->upsample) {
    ROMA_REQUIRE(cfg.upsample_h > 0, "roma_forward: upsample pass on a handle without an upsample resolution");
    ROMA_REQUIRE(fw->seed_flow && fw->seed_cert && fw->seed_h > 0 && fw->seed_w > 0,
                 "roma_forward: the upsample pass needs batch[\"corresps\"] (seed_flow / seed_cert)");
  }
  ROMA_CHECK_HIP(hipSetDevice(cfg.device));
  return match_impl(B, ima, imb, nullptr, nullptr, nullptr, nullptr, st, false, arena, persist, fw);
}

int Model::match_streams(int B, const float* ima, const float* imb, const float* ima_hr, const float* imb_hr, float* warp,
                         float* cert, hipStream_t st) {
  static const int env_streams = getenv("ROMA_STREAMS") ? atoi(getenv("ROMA_STREAMS")) : 0;
  static const bool serial_env = getenv("ROMA_STREAMS_SERIAL") && atoi(getenv("ROMA_STREAMS_SERIAL")) != 0;
  const int ns = (debug && g_dbg_dual_slot < 0) ? 1 : std::min(std::min(env_streams > 0 ? env_streams : n_streams, (int)MAX_STREAMS), B);
  if (ns <= 1) return match_impl(B, ima, imb, ima_hr, imb_hr, warp, cert, st, false, arena, persist);
  if (int rc = ensure_side_streams(ns)) return rc;
  // fork: the side streams start after everything already queued on the caller's stream (inputs); join at the end
  const size_t im_lo = (size_t)3 * cfg.coarse_h * cfg.coarse_w, im_hi = (size_t)3 * cfg.upsample_h * cfg.upsample_w;
  const int Ho = cfg.upsample_preds ? cfg.upsample_h : cfg.coarse_h, Wo = cfg.upsample_preds ? cfg.upsample_w : cfg.coarse_w;
  const size_t px = (size_t)Ho * Wo * (cfg.symmetric ? 2 : 1);
  ROMA_CHECK_HIP(hipEventRecord(ev_fork, st));
  int rc = 0, b0 = 0;
  for (int i = 0; i < ns; ++i) {  // balanced contiguous split; part 0 (the largest) stays on the caller's stream
    const int bi = B / ns + (i < B % ns ? 1 : 0);
    hipStream_t si = i == 0 ? st : side[i - 1];
    if (i > 0) ROMA_CHECK_HIP(hipStreamWaitEvent(si, ev_fork, 0));
    if (i > 0 && serial_env) {  // diagnostic: same streams and arenas, but no overlap - sub-batch i starts after i - 1 ended
      hipStream_t sp = i == 1 ? st : side[i - 2];
      ROMA_CHECK_HIP(hipEventRecord(ev_join[i - 1], sp));
      ROMA_CHECK_HIP(hipStreamWaitEvent(si, ev_join[i - 1], 0));
    }
    if (!rc)
      rc = match_impl(bi, ima + b0 * im_lo, imb + b0 * im_lo, ima_hr ? ima_hr + b0 * im_hi : nullptr,
                      imb_hr ? imb_hr + b0 * im_hi : nullptr, warp + b0 * px * 4, cert + b0 * px, si, false,
                      i == 0 ? arena : side_arena[i - 1], i == 0 ? persist : side_persist[i - 1]);
    b0 += bi;
  }
  // always join, also on error: the caller's stream must not run ahead of work already queued on a side stream
  for (int i = 1; i < ns; ++i) {
    (void)hipEventRecord(ev_join[i - 1], side[i - 1]);
    (void)hipStreamWaitEvent(st, ev_join[i - 1], 0);
  }
  return rc;
}

// ---------------------------------------------------------------- blocked Cholesky solve (transposed RHS)
// Solves A X = F for SPD A (n x n, n % 64 == 0) with Rt = F^T [d x n] overwritten by X^T.
// Right-looking 64-wide blocked factorisation; diagonal blocks are factorised and explicitly inverted by one
// workgroup (chol_diag), everything else is the MFMA GEMM with alpha = -1 accumulate:
//   panel  L[i,k]  = A[i,k] Linv_kk^T ; trailing A[i,j] -= L[i,k] L[j,k]^T (lower tiles only)
//   fwd    Yt[:,k] = Rt[:,k] Linv_kk^T ; Rt[:,i>k] -= Yt[:,k] L[i,k]^T
//   bwd    Xt[:,k] = Rt[:,k] Linv_kk   ; Rt[:,j<k] -= Xt[:,k] L[k,j]      (via LT = L^T)
// Round 4 - the augmented form.  The forward substitution does to the d rows of Rt exactly what the factorisation does to the
// rows of A below the diagonal block: multiply the column block k by Linv_kk^T, then subtract its product with L[j,k]^T from the
// column blocks j > k.  When the caller stores Rt right behind A (one (n + d) x n matrix per item), the panel and the trailing
// GEMM of step k simply run over mrem + d rows and the 2 * nblk - 1 launches of the forward loop disappear from the GP's
// launch-latency-bound chain (~0.75 ms of 10 .. 25 us launches at n = 1600); same operations on every element in the same
// order, so the result is bit-identical to the separate loop.
int g_pool_proj = -1;  // roma_tuning("pool_proj", v): 1 = max-pool + proj head of strides 1 / 2 in one pass (default), 0 = separate kernels, -1 = env ROMA_POOL_PROJ
int g_gp_col = -1;  // roma_tuning("gp_col", v): 1 = left-looking block-column kernel (default), 0 = right-looking chain, -1 = env ROMA_GP_COL
int cholesky_solve_t(float* A, float* Rt, float* LT, float* Linv, float* LinvT, int n, int d, int batch, hipStream_t st,
                     long strideA, long strideR) {
  ROMA_REQUIRE(n % 64 == 0 && n > 0 && d % 4 == 0, "cholesky_solve: n must be a multiple of 64, d of 4");
  const int nblk = n / 64;
  const long sA = strideA > 0 ? strideA : (long)n * n, sR = strideR > 0 ? strideR : (long)d * n, sL = (long)nblk * 4096;
  const long sLT = (long)n * n;  // LT is always dense
  static const bool aug_env = !(getenv("ROMA_GP_AUG") && atoi(getenv("ROMA_GP_AUG")) == 0);  // A/B: 0 = always the separate forward loop
  const bool aug = aug_env && Rt == A + (long)n * n && (batch == 1 || (sA == sR && sA >= (long)(n + d) * n));
  const int extra = aug ? d : 0;
  // Round 6: the augmented system left-looking, ONE launch per block column (chol_col.hip) instead of chol_diag + panel +
  // trailing update; L^T is written by the same launches.  roma_tuning("gp_col", 0) / ROMA_GP_COL=0: the right-looking chain.
  static const bool col_env = !(getenv("ROMA_GP_COL") && atoi(getenv("ROMA_GP_COL")) == 0);
  const bool col = aug && (g_gp_col >= 0 ? g_gp_col != 0 : col_env) && d >= 64 && d % 64 == 0;
  static std::atomic<unsigned> solve_epoch{0};  // tags the in-launch hand-off flags of this solve (chol_col.hip)
  const unsigned epoch = col ? ++solve_epoch : 0u;
  for (int k = 0; k < nblk && col; ++k)
    if (int rc = chol_col_launch(A, n, sA, LT, sLT, n, d, Linv, LinvT, k, nblk, batch, epoch, st)) return rc;
  if (col)
    if (int rc = chol_col_restore_launch(A, n, sA, LT, sLT, n, batch, st)) return rc;
  for (int k = 0; k < nblk && !col; ++k) {
    if (int rc = chol_diag_launch(A, n, sA, Linv, LinvT, k, nblk, batch, st)) return rc;
    const int mrem = n - (k + 1) * 64;
    if (mrem + extra <= 0) break;
    GemmArgs g;
    g.A = A + (long)(k + 1) * 64 * n + k * 64; g.lda = n; g.sA = sA;
    g.W = Linv + (long)k * 4096; g.ldw = 64; g.sW = sL;
    g.C = const_cast<float*>(static_cast<const float*>(g.A)); g.ldc = n; g.sC = sA;
    g.M = mrem + extra; g.N = 64; g.K = 64; g.batch = batch;
    if (int rc = gemm_launch(g, st)) return rc;
    if (mrem <= 0) break;
    GemmArgs t;
    t.A = g.A; t.lda = n; t.sA = sA;
    t.W = g.A; t.ldw = n; t.sW = sA;
    float* Ct = A + (long)(k + 1) * 64 * n + (k + 1) * 64;
    t.C = Ct; t.ldc = n; t.sC = sA;
    t.res = Ct; t.ldr = n; t.sR = sA;
    t.alpha = -1.f; t.lower_only = 1;
    t.M = mrem + extra; t.N = mrem; t.K = 64; t.batch = batch;
    if (int rc = gemm_launch(t, st)) return rc;
  }
  if (!col)
    if (int rc = transpose_launch(A, LT, n, n, batch, st, sA, sLT)) return rc;
  for (int k = 0; k < nblk && !aug; ++k) {  // forward (separate right-hand sides only)
    GemmArgs g;
    g.A = Rt + k * 64; g.lda = n; g.sA = sR;
    g.W = Linv + (long)k * 4096; g.ldw = 64; g.sW = sL;
    g.C = Rt + k * 64; g.ldc = n; g.sC = sR;
    g.M = d; g.N = 64; g.K = 64; g.batch = batch;
    if (int rc = gemm_launch(g, st)) return rc;
    const int mrem = n - (k + 1) * 64;
    if (mrem <= 0) break;
    GemmArgs u;
    u.A = Rt + k * 64; u.lda = n; u.sA = sR;
    u.W = A + (long)(k + 1) * 64 * n + k * 64; u.ldw = n; u.sW = sA;
    u.C = Rt + (k + 1) * 64; u.ldc = n; u.sC = sR;
    u.res = Rt + (k + 1) * 64; u.ldr = n; u.sR = sR;
    u.alpha = -1.f;
    u.M = d; u.N = mrem; u.K = 64; u.batch = batch;
    if (int rc = gemm_launch(u, st)) return rc;
  }
  static const bool bwd_env = !(getenv("ROMA_GP_BWD2") && atoi(getenv("ROMA_GP_BWD2")) == 0);  // A/B: 0 = two launches per backward step
  if (bwd_env && nblk > 1) {
    // Backward substitution with ONE launch per step.  X_k = R_k Linv_kk and R_j -= X_k L[k,j] (j < k) re-associate to
    // R_j -= R_k (Linv_kk L[k,j]): the products M_k = Linv_kk L[k, :k] do not depend on the right-hand sides, so they are formed
    // for all k at once (in place over L^T: LT[:, block k] <- LT[:, block k] LinvT_kk, one launch with the second batch
    // level over k), the chain is the nblk - 1 update GEMMs on the not-yet-scaled R_k, and all X_k = R_k Linv_kk follow in one
    // launch at the end.  2 + nblk - 1 launches instead of 2 nblk - 1 on the GP's launch-latency-bound chain.
    GemmArgs m;
    m.A = LT; m.lda = n; m.sA = sLT; m.sA2 = 64;
    m.W = Linv; m.ldw = 64; m.sW = sL; m.sW2 = 4096;
    m.C = LT; m.ldc = n; m.sC = sLT; m.sC2 = 64;
    m.M = n; m.N = 64; m.K = 64; m.batch = batch; m.batch2 = nblk;
    if (int rc = gemm_launch(m, st)) return rc;
    for (int k = nblk - 1; k >= 1; --k) {
      GemmArgs u;
      u.A = Rt + k * 64; u.lda = n; u.sA = sR;
      u.W = LT + k * 64; u.ldw = n; u.sW = sLT;
      u.C = Rt; u.ldc = n; u.sC = sR;
      u.res = Rt; u.ldr = n; u.sR = sR;
      u.alpha = -1.f;
      u.M = d; u.N = k * 64; u.K = 64; u.batch = batch;
      if (int rc = gemm_launch(u, st)) return rc;
    }
    GemmArgs x;
    x.A = Rt; x.lda = n; x.sA = sR; x.sA2 = 64;
    x.W = LinvT; x.ldw = 64; x.sW = sL; x.sW2 = 4096;
    x.C = Rt; x.ldc = n; x.sC = sR; x.sC2 = 64;
    x.M = d; x.N = 64; x.K = 64; x.batch = batch; x.batch2 = nblk;
    return gemm_launch(x, st);
  }
  for (int k = nblk - 1; k >= 0; --k) {  // backward
    GemmArgs g;
    g.A = Rt + k * 64; g.lda = n; g.sA = sR;
    g.W = LinvT + (long)k * 4096; g.ldw = 64; g.sW = sL;
    g.C = Rt + k * 64; g.ldc = n; g.sC = sR;
    g.M = d; g.N = 64; g.K = 64; g.batch = batch;
    if (int rc = gemm_launch(g, st)) return rc;
    if (k == 0) break;
    GemmArgs u;
    u.A = Rt + k * 64; u.lda = n; u.sA = sR;
    u.W = LT + k * 64; u.ldw = n; u.sW = sLT;
    u.C = Rt; u.ldc = n; u.sC = sR;
    u.res = Rt; u.ldr = n; u.sR = sR;
    u.alpha = -1.f;
    u.M = d; u.N = k * 64; u.K = 64; u.batch = batch;
    if (int rc = gemm_launch(u, st)) return rc;
  }
  return 0;
}

// ---------------------------------------------------------------- GP match encoder (matcher.py:291-323, 191-200, 274-289)
// pf: projected stride-16 features of the 2B images [2B, n, ldf] (n = th * tw tokens, 512 channels; images [0, B) = A,
// [B, 2B) = B).  Directed pair i (i < B, or i < 2B when symmetric) has query image i and support image (i + B) % 2B.
// mu[i, :, 0:512] = K_xy (K_yy + 0.1 I)^-1 cos(8 pi (W_pos grid + b_pos)), written with row stride ld_mu (f32).
// K_yy, its Cholesky factor and alpha depend only on the SUPPORT image, so they are computed once per image.
#define GP_RUN(expr)              \
  do {                            \
    if (!dry) {                   \
      int _rc = (expr);           \
      if (_rc) return _rc;        \
    }                             \
  } while (0)
int gp_posterior(const void* pf, long ldf, int act_dt, int B, bool symmetric, int th, int tw, const float* gp_w,
                 const float* gp_b, float* mu, long ld_mu, Arena& arena, hipStream_t st, bool dry) {
  const size_t esz = act_dt == DT_F32 ? 4 : 2;
  const int nimg = 2 * B, ndp = symmetric ? 2 * B : B, shift = B;
  const int n = th * tw, npad = (int)round_up(n, 64), nblk = npad / 64;
  auto AL = [&](size_t elems, size_t es) { return arena.alloc(elems * es); };
  auto off = [&](const void* p, long elems) -> const void* { return static_cast<const char*>(p) + elems * (long)esz; };
  float* norms = (float*)AL((size_t)nimg * n, 4);
  // K_yy and the right-hand sides F^T of an image in ONE (npad + 512) x npad matrix: cholesky_solve_t then runs the forward
  // substitution inside the factorisation loop
  const long saug = (long)(npad + 512) * npad;
  float* Kyy = (float*)AL((size_t)nimg * saug, 4);
  float* Rt = Kyy + (long)npad * npad;  // image j: Kyy + j * saug, Rt + j * saug
  float* LT = (float*)AL((size_t)nimg * npad * npad, 4);
  float* Kxy = (float*)AL((size_t)ndp * n * npad, 4);
  float* Linv = (float*)AL((size_t)nimg * nblk * 4096, 4);
  float* LinvT = (float*)AL((size_t)nimg * nblk * 4096, 4);
  GP_RUN(rownorm_launch(pf, ldf, act_dt, norms, (long)nimg * n, 512, st));
  // support images actually needed: symmetric -> all, else images [B, 2B)
  const int j0 = symmetric ? 0 : B, nj = symmetric ? nimg : B;
  {
    GemmArgs g;  // K_yy + sigma^2 I  (cosine kernel, CosKernel matcher.py:191-200)
    g.A = off(pf, (long)j0 * n * ldf); g.lda = ldf; g.sA = (long)n * ldf;
    g.W = g.A; g.ldw = ldf; g.sW = g.sA;
    g.C = Kyy + (long)j0 * saug; g.ldc = npad; g.sC = saug;
    g.M = n; g.N = n; g.K = 512; g.batch = nj; g.in_dt = act_dt; g.out_dt = DT_F32; g.mode = EPI_COSK;
    g.nx = norms + (long)j0 * n; g.ny = g.nx; g.sNx = n; g.sNy = n; g.inv_t = 1.0f / 0.2f; g.diag_add = 0.1f;
    GP_RUN(gemm_launch(g, st));
  }
  GP_RUN(pad_identity_launch(Kyy + (long)j0 * saug, npad, saug, n, npad, nj, st));
  if (!dry && npad != n) ROMA_CHECK_HIP(hipMemsetAsync(Kxy, 0, (size_t)ndp * n * npad * 4, st));  // (pad columns only: the GEMM writes the rest)
  for (int half = 0; half < (symmetric ? 2 : 1); ++half) {
    GemmArgs g;  // K_xy for directed pairs [half*B, half*B + B): x = image i, y = image (i + B) % nimg
    const int i0 = half * B, s0 = (i0 + shift) % nimg;
    g.A = off(pf, (long)i0 * n * ldf); g.lda = ldf; g.sA = (long)n * ldf;
    g.W = off(pf, (long)s0 * n * ldf); g.ldw = ldf; g.sW = (long)n * ldf;
    g.C = Kxy + (long)i0 * n * npad; g.ldc = npad; g.sC = (long)n * npad;
    g.M = n; g.N = n; g.K = 512; g.batch = B; g.in_dt = act_dt; g.out_dt = DT_F32; g.mode = EPI_COSK;
    g.nx = norms + (long)i0 * n; g.ny = norms + (long)s0 * n; g.sNx = n; g.sNy = n; g.inv_t = 1.0f / 0.2f;
    GP_RUN(gemm_launch(g, st));
  }
  GP_RUN(gp_basis_launch(gp_w, gp_b, Rt + (long)j0 * saug, 512, th, tw, npad, st, nj, saug));  // F^T behind every image's K_yy
  GP_RUN(cholesky_solve_t(Kyy + (long)j0 * saug, Rt + (long)j0 * saug, LT + (long)j0 * npad * npad,
                          Linv + (long)j0 * nblk * 4096, LinvT + (long)j0 * nblk * 4096, npad, 512, nj, st, saug, saug));
  for (int half = 0; half < (symmetric ? 2 : 1); ++half) {
    GemmArgs g;  // mu = K_xy alpha
    const int i0 = half * B, s0 = (i0 + shift) % nimg;
    g.A = Kxy + (long)i0 * n * npad; g.lda = npad; g.sA = (long)n * npad;
    g.W = Rt + (long)s0 * saug; g.ldw = npad; g.sW = saug;
    g.C = mu + (long)i0 * n * ld_mu; g.ldc = ld_mu; g.sC = (long)n * ld_mu;
    g.M = n; g.N = 512; g.K = npad; g.batch = B;
    GP_RUN(gemm_launch(g, st));
  }
  return 0;
}
#undef GP_RUN

// ---------------------------------------------------------------- the match() schedule
#define RUN(expr)                 \
  do {                            \
    if (!dry) {                   \
      int _rc = (expr);           \
      if (_rc) return _rc;        \
    }                             \
  } while (0)

int Model::match_impl(int B, const float* ima, const float* imb, const float* ima_hr, const float* imb_hr,
                      float* warp_out, float* cert_out, hipStream_t st, bool dry, Arena& arena, Arena& persist,
                      const roma_forward_args_t* fw) {
  const size_t esz = act_dt == DT_F32 ? 4 : 2;
  const int nimg = 2 * B;
  const bool sym = fw ? fw->symmetric != 0 : cfg.symmetric != 0;
  const int ndp = sym ? 2 * B : B;
  const int shift = B;  // support image of directed pair i = (i + B) % nimg
  arena.reset();
  persist.reset();
  auto AL = [&](size_t elems, size_t es) { return arena.alloc(elems * es); };
  // determinism trace: checksum of a stage's output, XORed into this sub-batch stream's table
  const int tslot = (&arena == &this->arena) ? 0 : (int)(&arena - side_arena) + 1;
  dbg_cur_slot = tslot;  // (per handle: two handles may be in match() from two threads)
  const bool tracing = trace_on && !dry;
  if (tracing) {
    if (!trace_dev[tslot]) {
      ROMA_CHECK_HIP(hipMalloc((void**)&trace_dev[tslot], TRACE_MAX * sizeof(unsigned long long)));
      owned.push_back(trace_dev[tslot]);
    }
    ROMA_CHECK_HIP(hipMemsetAsync(trace_dev[tslot], 0, TRACE_MAX * sizeof(unsigned long long), st));
    trace_n[tslot] = 0;
  }
  auto CK = [&](const std::string& name, const void* p, size_t bytes) -> int {
    if (!tracing) return 0;
    if (trace_n[tslot] >= TRACE_MAX) return 0;  // table full: later stages are not traced (roma_debug_trace sees <= TRACE_MAX)
    const int k = trace_n[tslot]++;
    if ((int)trace_names[tslot].size() <= k) trace_names[tslot].push_back(name);
    else trace_names[tslot][k] = name;
    return checksum_launch(p, bytes, trace_dev[tslot] + k, st);
  };
  auto off = [&](void* p, long elems) -> void* { return static_cast<char*>(p) + elems * (long)esz; };

  // ---- persistent, zero-initialised attention workspaces (pads must stay finite)
  const int th = cfg.coarse_h / 14, tw = cfg.coarse_w / 14, T = th * tw;
  const int Nd = T + 1, Npd = (int)round_up(Nd, 128), Npt = (int)round_up(T, 128);
  const size_t qkv_elems = std::max((size_t)nimg * 16 * Npd * 64, (size_t)ndp * 8 * Npt * 128);
  void* qbuf = persist.alloc(qkv_elems * esz);
  void* kbuf = persist.alloc(qkv_elems * esz);
  void* vtbuf = persist.alloc(qkv_elems * esz);

  // ---- results that live across the two passes
  float* cert16_keep = (float*)AL((size_t)ndp * T, 4);
  // The flow / certainty ping-pong buffers of BOTH passes live outside the per-pass scratch: the coarse pass's finest
  // correspondences seed the upsample pass and the last pass's feed the epilogue where they are - round 5: no device-to-device
  // copies at the end of a pass (2 x 72 MB + 2 x 30 MB per sub-batch and call before).
  const int Hfin = cfg.upsample_preds ? cfg.upsample_h : cfg.coarse_h;
  const int Wfin = cfg.upsample_preds ? cfg.upsample_w : cfg.coarse_w;
  float *pp_flow[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *pp_cert[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  for (int ps = 0; ps < ((cfg.upsample_preds || (fw && fw->upsample)) ? 2 : 1); ++ps) {
    const size_t px = (size_t)ndp * (ps ? cfg.upsample_h : cfg.coarse_h) * (ps ? cfg.upsample_w : cfg.coarse_w);
    for (int i = 0; i < 2; ++i) {
      pp_flow[ps][i] = (float*)AL(px * 2, 4);
      pp_cert[ps][i] = (float*)AL(px, 4);
    }
  }
  float *flow_p1 = nullptr, *cert_p1 = nullptr;    // finest correspondences of the coarse pass (wherever the ping-pong ended)
  float *flow_fin = nullptr, *cert_fin = nullptr;  // ... of the last pass

  // shared ViT block (vit.hip): DINOv2 goes through vit_forward / roma_vit_forward below, the coordinate decoder calls the
  // block directly (f32 residual stream)
  auto vit_block = [&](const VitBlockW& w, void* x, int x_dt, long rows, int Bn, int N, int npad, int heads, int hd, float eps,
                       void* ln, void* ao, void* hid) -> int {
    VitScratch sc;
    sc.ln = ln; sc.ao = ao; sc.hid = hid; sc.q = qbuf; sc.k = kbuf; sc.vt = vtbuf;
    RUN(vit_block_run(w.pub(), x, x_dt, rows, Bn, N, npad, heads, hd, eps, act_dt, sc, st));
    return 0;
  };

  const int pass0 = fw ? (fw->upsample ? 1 : 0) : 0, pass1 = fw ? pass0 + 1 : (cfg.upsample_preds ? 2 : 1);
  for (int pass = pass0; pass < pass1; ++pass) {
    const bool up = pass == 1;
    const int H = up ? cfg.upsample_h : cfg.coarse_h, W = up ? cfg.upsample_w : cfg.coarse_w;
    const float* imA = (up && !fw) ? ima_hr : ima;  // roma_forward hands the pass's own images over as im_a / im_b
    const float* imB = (up && !fw) ? imb_hr : imb;
    const size_t pass_mark = arena.mark();
    // =============================== encoder: VGG19-BN pyramid (encoders.py:17-27)
    void* feat[5] = {nullptr};  // index by log2(stride): 0 -> stride 1 ... 3 -> stride 8 ; [4] = stride 16 (DINOv2)
    const int fh[4] = {H, H / 2, H / 4, H / 8}, fw_[4] = {W, W / 2, W / 4, W / 8};
    const int fc[4] = {64, 128, 256, 512};
    for (int l = 0; l < 4; ++l) feat[l] = AL((size_t)nimg * fh[l] * fw_[l] * fc[l], esz);
    // Round 6 (pool_proj.hip): at strides 1 and 2 the max-pool and the proj head of the level read the un-pooled map in ONE
    // pass (16-bit modes); the projected maps are then ready when the decoder reaches those scales.  pf_pre[l]: stride 2^l.
    static const bool pp_env = !(getenv("ROMA_POOL_PROJ") && atoi(getenv("ROMA_POOL_PROJ")) == 0);
    void* pf_pre[2] = {nullptr, nullptr};
    const int pp_si[2] = {4, 3};  // index of stride 1 / 2 in SCALES
    bool pp_on[2];
    for (int l = 0; l < 2; ++l) {
      const RefinerW& rr = ref[pp_si[l]];
      const int ldf_l = (int)round_up(rr.Cf, 8);
      pp_on[l] = (g_pool_proj >= 0 ? g_pool_proj != 0 : pp_env) && fh[l] >= 2 && fw_[l] >= 2 &&
                 pool_proj_supported(fc[l], rr.Cf, ldf_l, act_dt) && proj[pp_si[l]].b != nullptr;
      if (pp_on[l] || dry) pf_pre[l] = AL((size_t)nimg * fh[l] * fw_[l] * ldf_l, esz);  // (planned whatever the switch says now)
      if (!pp_on[l] && !dry) pf_pre[l] = nullptr;
    }
    {
      const size_t enc_mark = arena.mark();
      void* t0 = AL((size_t)nimg * H * W * 64, esz);
      void* t1 = AL((size_t)nimg * (H / 2) * (W / 2) * 64, esz);
      static const int c64_env = getenv("ROMA_CONV64") ? atoi(getenv("ROMA_CONV64")) : 7;
      void* col1 = AL((size_t)nimg * H * W * 32, esz);  // (planned whichever form runs: the switch may change between calls)
      if (act_dt == DT_BF16 && ((g_conv64_mode >= 0 ? g_conv64_mode : c64_env) & 4) && (long)3 * H * W < (1l << 23)) {  // (its packed tap offsets)
        // first layer (Cin = 3), bf16: fused kernel straight from the f32 image (conv64.hip)
        RUN(conv3x3_c3_bf16_launch(imA, vgg[0].w, vgg[0].b, t0, B, H, W, st));
        RUN(conv3x3_c3_bf16_launch(imB, vgg[0].w, vgg[0].b, off(t0, (long)B * H * W * 64), B, H, W, st));
      } else {  // im2col to K = 32, then the same MFMA GEMM as every other layer
        RUN(im2col3x3_c3_launch(imA, col1, B, H, W, act_dt, st));
        RUN(im2col3x3_c3_launch(imB, off(col1, (long)B * H * W * 32), B, H, W, act_dt, st));
        GemmArgs g;
        g.A = col1; g.lda = 32; g.W = vgg[0].w; g.ldw = vgg[0].ldw; g.C = t0; g.ldc = 64;
        g.M = nimg * H * W; g.N = 64; g.K = 32; g.k_alg = 27; g.in_dt = act_dt; g.out_dt = act_dt; g.bias = vgg[0].b; g.act = ACT_RELU;
        RUN(gemm_launch(g, st));
      }
      auto conv = [&](int li, const void* in, void* out, int h, int w) -> int {
        GemmArgs g;
        g.A = in; g.W = vgg[li].w; g.ldw = vgg[li].ldw; g.C = out; g.ldc = vgg_cout[li];
        g.M = nimg * h * w; g.N = vgg_cout[li]; g.K = 9 * vgg_cin[li];
        g.in_dt = act_dt; g.out_dt = act_dt; g.bias = vgg[li].b; g.act = ACT_RELU;
        g.conv_h = h; g.conv_w = w; g.conv_c = vgg_cin[li]; g.conv_korder = vgg_korder[li];
        RUN(gemm_launch(g, st));
        return 0;
      };
      auto pool = [&](int l, const void* in, void* out) -> int {
        if (pp_on[l]) {
          const int si_ = pp_si[l];
          RUN(pool_proj_launch(in, out, pf_pre[l], proj[si_].w, proj[si_].ldw, proj[si_].b, ref[si_].Cf, (int)round_up(ref[si_].Cf, 8),
                               nimg, fh[l], fw_[l], fc[l], act_dt, st));
        } else {
          RUN(maxpool2x2_launch(in, out, nimg, fh[l], fw_[l], fc[l], act_dt, st));
        }
        return 0;
      };
      if (int rc = conv(1, t0, feat[0], H, W)) return rc;
      if (int rc = pool(0, feat[0], t1)) return rc;
      if (int rc = conv(2, t1, t0, H / 2, W / 2)) return rc;
      if (int rc = conv(3, t0, feat[1], H / 2, W / 2)) return rc;
      if (int rc = pool(1, feat[1], t1)) return rc;
      if (int rc = conv(4, t1, t0, H / 4, W / 4)) return rc;
      if (int rc = conv(5, t0, t1, H / 4, W / 4)) return rc;
      if (int rc = conv(6, t1, t0, H / 4, W / 4)) return rc;
      if (int rc = conv(7, t0, feat[2], H / 4, W / 4)) return rc;
      RUN(maxpool2x2_launch(feat[2], t1, nimg, H / 4, W / 4, 256, act_dt, st));
      if (int rc = conv(8, t1, t0, H / 8, W / 8)) return rc;
      if (int rc = conv(9, t0, t1, H / 8, W / 8)) return rc;
      if (int rc = conv(10, t1, t0, H / 8, W / 8)) return rc;
      if (int rc = conv(11, t0, feat[3], H / 8, W / 8)) return rc;
      arena.release(enc_mark);
    }
    for (int l = 0; l < 4; ++l)
      if (int rc = CK((up ? "p2_feat" : "p1_feat") + std::to_string(1 << l), feat[l], (size_t)nimg * fh[l] * fw_[l] * fc[l] * esz)) return rc;
    if (fw && !dry)  // extract_backbone_features (matcher.py:585-596): fw->feat[] is ordered 16, 8, 4, 2, 1
      for (int l = 0; l < 4; ++l)
        if (fw->feat[4 - l])
          ROMA_CHECK_HIP(hipMemcpyAsync(fw->feat[4 - l], feat[l], (size_t)nimg * fh[l] * fw_[l] * fc[l] * esz, hipMemcpyDeviceToDevice, st));
    if (debug && !dry && !up) {
      for (int l = 0; l < 4; ++l) {
        const std::string nm = "feat" + std::to_string(1 << l);
        if (int rc = dbg_save(nm.c_str(), feat[l], (size_t)nimg * fh[l] * fw_[l] * fc[l] * esz, st)) return rc;
      }
    }
    // shared transformer scratch (DINOv2 rows >= decoder rows)
    const long rows_d = (long)nimg * Nd, rows_t = (long)ndp * T;
    void *ln = nullptr, *ao = nullptr, *hid = nullptr;
    if (!up) {
      feat[4] = AL((size_t)nimg * T * 1024, esz);
      const long rmax = std::max(rows_d, rows_t);
      ln = AL((size_t)rmax * 1024, esz);
      ao = AL((size_t)rmax * 1024, esz);
      hid = AL((size_t)rmax * 4096, esz);
      // =============================== encoder: DINOv2 ViT-L/14 (dinov2.py:192-237)
      const size_t dmark = arena.mark();
      void* col = AL((size_t)nimg * T * patch.ldw, esz);
      float* pt = (float*)AL((size_t)nimg * T * 1024, 4);
      float* x = (float*)AL((size_t)rows_d * 1024, 4);
      static const bool res_f32_env = getenv("ROMA_VIT_RES_F32") && atoi(getenv("ROMA_VIT_RES_F32")) != 0;
      void* xs = nullptr;
      if (act_dt == DT_BF16) xs = AL((size_t)rows_d * 1024, 2);  // 16-bit residual stream: always planned, the option may flip later
      void* feat_vit = feat[4];
      if (mixed) feat_vit = AL((size_t)nimg * T * 1024, 2);       // bfloat16 patch tokens of the sibling library
      if (!dry) {
        roma_vit_block_t blk[24];
        for (int i = 0; i < 24; ++i) blk[i] = dino[i].pub();
        roma_vit_args_t va{};
        va.B = B; va.H = H; va.W = W;
        va.act = act_dt == DT_F32 ? ROMA_F32 : (mixed ? ROMA_BF16 : roma_h16_format());
        va.bf16_residual = act_dt == DT_BF16 && vit_bf16_residual && !res_f32_env;
        va.im_a = imA; va.im_b = imB;
        va.patch_w = patch.w; va.patch_b = patch.b; va.patch_ldw = patch.ldw;
        va.cls_tok = cls_tok; va.pos_emb = pos_emb;
        va.blocks = blk; va.nblocks = 24;
        va.norm_w = dino_nw; va.norm_b = dino_nb;
        va.col = col; va.pt = pt; va.x = x; va.xs = xs; va.ln = ln; va.ao = ao; va.hid = hid;
        va.q = qbuf; va.k = kbuf; va.vt = vtbuf;
        va.feat_out = feat_vit;
        if (mixed) {
          // DINOv2 in bfloat16 (the reference's amp_dtype reaches only it, roma_models.py:183-188) in the bf16 build of this
          // library, on this sub-batch's stream; autocast then casts its features to binary16 for the proj head
          if (int rc = peer_vit_forward(&va, st)) {
            set_error("ROMA_MIXED: roma_vit_forward of the bfloat16 library failed");
            return rc;
          }
          RUN(convert_from_bf16_launch(feat_vit, feat[4], (long)nimg * T * 1024, st));
        } else {
          RUN(vit_forward(va, st));
        }
      }
      arena.release(dmark);
      if (int rc = CK("feat16", feat[4], (size_t)nimg * T * 1024 * esz)) return rc;
      if (debug && !dry)
        if (int rc = dbg_save("feat16", feat[4], (size_t)nimg * T * 1024 * esz, st)) return rc;
      if (fw && !dry && fw->feat[0])
        ROMA_CHECK_HIP(hipMemcpyAsync(fw->feat[0], feat[4], (size_t)nimg * T * 1024 * esz, hipMemcpyDeviceToDevice, st));
    }

    // =============================== decoder (matcher.py:395-527)
    float *flow = pp_flow[pass][0], *cert = pp_cert[pass][0], *flow_alt = pp_flow[pass][1], *cert_alt = pp_cert[pass][1];
    int ch = 0, cw = 0;  // current flow map size
    if (up) {
      ch = H / 8; cw = W / 8;
      if (fw) {  // batch["corresps"] of the caller, any resolution (matcher.py:423-435)
        RUN(resize_bilinear_launch(fw->seed_flow, flow, ndp, fw->seed_h, fw->seed_w, ch, cw, 2, st));
        RUN(resize_bilinear_launch(fw->seed_cert, cert, ndp, fw->seed_h, fw->seed_w, ch, cw, 1, st));
      } else {
        RUN(resize_bilinear_launch(flow_p1, flow, ndp, cfg.coarse_h, cfg.coarse_w, ch, cw, 2, st));
        RUN(resize_bilinear_launch(cert_p1, cert, ndp, cfg.coarse_h, cfg.coarse_w, ch, cw, 1, st));
      }
    }
    double scale_factor = sqrt((double)H * (double)W / (560.0 * 560.0));  // matcher.py:805, 877-881
    if (!up && coarse_scale_factor > 0.0) scale_factor = coarse_scale_factor;
    if (fw && fw->scale_factor > 0.0) scale_factor = fw->scale_factor;
    for (int si = up ? 1 : 0; si < 5; ++si) {
      const int ins = SCALE_INT[si];
      const int hs = ins == 16 ? th : H / ins, ws = ins == 16 ? tw : W / ins;
      const long hw = (long)hs * ws;
      const RefinerW& r = ref[si];
      const size_t smark = arena.mark();
      // ---- proj head, once per image (proj(f_s) == swap(proj(f_q)) since it is per-image)
      const int ldf = (int)round_up(r.Cf, 8);
      const int lvl = ins == 16 ? 4 : (ins == 8 ? 3 : (ins == 4 ? 2 : (ins == 2 ? 1 : 0)));
      const bool pf_ready = lvl < 2 && pp_on[lvl];  // projected with the max-pool of the level (pool_proj.hip)
      void* pf = pf_ready ? pf_pre[lvl] : AL((size_t)nimg * hw * ldf, esz);
      if (!pf_ready) {
        GemmArgs g;
        g.A = feat[lvl]; g.lda = PROJ_CIN[si]; g.W = proj[si].w; g.ldw = proj[si].ldw; g.C = pf; g.ldc = ldf;
        g.M = (int)(nimg * hw); g.N = r.Cf; g.K = PROJ_CIN[si]; g.in_dt = act_dt; g.out_dt = act_dt; g.bias = proj[si].b;
        RUN(gemm_launch(g, st));
      }
      const std::string tp = std::string(up ? "p2_s" : "p1_s") + SCALES[si];
      if (int rc = CK(tp + "_proj", pf, (size_t)nimg * hw * ldf * esz)) return rc;
      if (ins == 16) {
        if (debug && !dry)
          if (int rc = dbg_save("proj16", pf, (size_t)nimg * hw * ldf * esz, st)) return rc;
        // ================= GP match encoder (matcher.py:291-323), f32
        const size_t gmark = arena.mark();
        float* tokens = (float*)AL((size_t)rows_t * 1024, 4);
        const size_t gmark2 = arena.mark();
        if (int rc = gp_posterior(pf, ldf, act_dt, B, sym, th, tw, gp_w, gp_b, tokens, 1024, arena, st, dry)) return rc;
        arena.release(gmark2);
        RUN(copy2d_launch(pf, ldf, act_dt, tokens + 512, 1024, DT_F32, rows_t, 512, st));
        if (debug && !dry)
          if (int rc = dbg_save("tokens16", tokens, (size_t)rows_t * 1024 * 4, st)) return rc;
        // ================= coordinate decoder (transformer/__init__.py:30-46)
        for (int i = 0; i < 5; ++i)
          if (int rc = vit_block(tdec[i], tokens, DT_F32, rows_t, ndp, T, Npt, 8, 128, 1e-5f, ln, ao, hid)) return rc;
        const int ldl = 4104;
        float* logits = (float*)AL((size_t)rows_t * ldl, 4);
        const void* zin = tokens;
        if (act_dt != DT_F32) {
          RUN(copy2d_launch(tokens, 1024, DT_F32, ln, 1024, act_dt, rows_t, 1024, st));
          zin = ln;
        }
        {
          GemmArgs g;
          g.A = zin; g.lda = 1024; g.W = to_out.w; g.ldw = to_out.ldw; g.C = logits; g.ldc = ldl;
          g.M = (int)rows_t; g.N = to_out.N; g.K = 1024; g.in_dt = act_dt; g.out_dt = DT_F32; g.bias = to_out.b;  // 4100: 4097 + 3 zero rows
          RUN(gemm_launch(g, st));
        }
        if (debug && !dry)
          if (int rc = dbg_save("logits16", logits, (size_t)rows_t * ldl * 4, st)) return rc;
        if (int rc = CK(tp + "_tokens_gp", tokens, (size_t)rows_t * 1024 * 4)) return rc;
        if (int rc = CK(tp + "_logits", logits, (size_t)rows_t * ldl * 4)) return rc;
        RUN(cls_to_flow_launch(logits, ldl, flow, cert, rows_t, st));
        if (int rc = CK(tp + "_gm_flow", flow, (size_t)rows_t * 2 * 4)) return rc;
        ch = th; cw = tw;
        if (debug && !dry) {  // tests: the computed coarse match first (so the flips can be counted), then the override
          if (int rc = dbg_save("gm_flow16_own", flow, (size_t)rows_t * 2 * 4, st)) return rc;
          auto inj = [&](const char* nm, float* dst, size_t bytes) -> int {
            auto it = inject.find(nm);
            if (it == inject.end()) return 0;
            ROMA_REQUIRE(it->second.second == bytes, "roma_debug_inject: injected stage has the wrong size for this batch");
            ROMA_CHECK_HIP(hipMemcpyAsync(dst, it->second.first, bytes, hipMemcpyDeviceToDevice, st));
            return 0;
          };
          if (int rc = inj("gm_flow16", flow, (size_t)rows_t * 2 * 4)) return rc;
          if (int rc = inj("gm_cert16", cert, (size_t)rows_t * 4)) return rc;
        }
        if (debug && !dry) {
          if (int rc = dbg_save("gm_flow16", flow, (size_t)rows_t * 2 * 4, st)) return rc;
          if (int rc = dbg_save("gm_cert16", cert, (size_t)rows_t * 4, st)) return rc;
        }
        arena.release(gmark);
        // pf must survive (allocated before gmark) - it does.
      }
      // ================= ConvRefiner (matcher.py:124-179)
      {
        const long M = (long)ndp * hw;
        void* d0 = AL((size_t)M * r.Cp, esz);
        void* d1 = AL((size_t)M * r.Cp, esz);
        RefinerInputArgs ia;
        ia.feat = pf; ia.ldf = ldf; ia.flow = flow; ia.d = d0; ia.ldd = r.Cp; ia.emb_w = r.emb_w; ia.emb_b = r.emb_b;
        ia.B = ndp; ia.H = hs; ia.W = ws; ia.C = r.Cf; ia.E = r.E; ia.Kcorr = r.K; ia.nimg = nimg; ia.shift = shift;
        ia.disp_scale = (float)(40.0 / 32.0 * scale_factor); ia.dt = act_dt;
        if (tracing)
          if (int rc = CK(tp + "_flow_before", flow, (size_t)M * 2 * 4)) return rc;
        RUN(refiner_input_launch(ia, st));
        if (r.radius) {
          LocalCorrArgs lc;
          lc.f0 = pf; lc.f1 = pf; lc.warp = flow; lc.out = off(d0, 2 * r.Cf + r.E);
          lc.B = ndp; lc.H = hs; lc.W = ws; lc.C = r.Cf; lc.radius = r.radius; lc.ld0 = ldf; lc.ld1 = ldf; lc.ldo = r.Cp;
          lc.nimg = nimg; lc.f1_shift = shift; lc.scale = 1.0f / sqrtf((float)r.Cf); lc.in_dt = act_dt; lc.out_dt = act_dt;
          const long lc_ints = local_corr_ws_ints(ndp, hs, ws, r.radius);  // tile work lists + bin tables + sorted query list (local_corr.h)
          lc.ws = (int*)AL((size_t)lc_ints, 4); lc.ws_bytes = lc_ints * 4;
          RUN(local_corr_window_launch(lc, st));
        }
        if (debug && !dry) {
          const std::string nm = std::string("p") + (up ? "2" : "1") + "_din" + SCALES[si];
          if ((size_t)M * r.Cp * esz <= ((size_t)(g_dbg_dual_slot >= 0 ? 512 : 64) << 20))
            if (int rc = dbg_save(nm.c_str(), d0, (size_t)M * r.Cp * esz, st)) return rc;
          if (g_dbg_dual_slot >= 0 && ins == 1)
            if (int rc = dbg_save((std::string("p") + (up ? "2" : "1") + "_flowin1").c_str(), flow, (size_t)M * 2 * 4, st)) return rc;
        }
        void *dcur = d0, *dalt = d1;
        if (int rc = CK(tp + "_din", d0, (size_t)M * r.Cp * esz)) return rc;
        if (tracing) {  // diagnostic re-reads: are the inputs of the stage still what they were, is d0 stable?
          if (int rc = CK(tp + "_proj_again", pf, (size_t)nimg * hw * ldf * esz)) return rc;
          if (int rc = CK(tp + "_flow_again", flow, (size_t)M * 2 * 4)) return rc;
          if (int rc = CK(tp + "_din_again", d0, (size_t)M * r.Cp * esz)) return rc;
          if (esz == 2 && r.E > 0) {  // which part of d deviates: x | x_hat | displacement embedding | rest; which directed pair
            auto CKC = [&](const std::string& name, const void* p, long rows, int c0, int c1) -> int {
              if (trace_n[tslot] >= TRACE_MAX) return 0;
              const int k = trace_n[tslot]++;
              if ((int)trace_names[tslot].size() <= k) trace_names[tslot].push_back(name);
              else trace_names[tslot][k] = name;
              return checksum_cols_launch(p, rows, r.Cp, c0, c1, trace_dev[tslot] + k, st);
            };
            if (int rc = CKC(tp + "_din_x", d0, M, 0, r.Cf)) return rc;
            if (int rc = CKC(tp + "_din_xhat", d0, M, r.Cf, 2 * r.Cf)) return rc;
            if (int rc = CKC(tp + "_din_emb", d0, M, 2 * r.Cf, 2 * r.Cf + r.E)) return rc;
            if (int rc = CKC(tp + "_din_rest", d0, M, 2 * r.Cf + r.E, r.Cp)) return rc;
            for (int b = 0; b < ndp; ++b)
              if (int rc = CKC(tp + "_din_pair" + std::to_string(b), off(d0, (long)b * hw * r.Cp), hw, 0, r.Cp)) return rc;
          }
        }
        const bool fused = fuse_refiner_blocks && refiner_block_supported(r.Cp, act_dt);
        // (Running the nine-block chain over GROUPS of pairs whose ping-pong buffers fit the 256 MiB Infinity Cache was
        // measured in round 3 and is slower: 97.4 ms/step whole batch, 98.6 / 99.3 / 101.3 with 400 / 200 / 100 MiB
        // groups - the smaller launches lose more than the cache hits win, profiles/r03_v1_ab_attn_map_refiner_groups.log.)
        bool composed = false, final_done = false;
        if (dry && fused) {  // plan the FINAL block's delta buffer whatever "compose_out_conv" is now: the option may change later
          const size_t mk = arena.mark();
          (void)AL((size_t)M * 4, 4);
          arena.release(mk);
        }
        for (int b = 0; b < 9; ++b) {
          if (b == 8 && compose_out_conv && fused) {
            // narrow scales: the FINAL form of the fused block - depthwise + the composed C -> 3 map on the MFMA, 16 bytes of
            // deltas per pixel instead of a block output, then one small pass that adds them to flow / certainty
            float* delta = (float*)AL((size_t)M * 4, 4);
            RUN(refiner_block_final_launch(dcur, delta, r.dw_w[b], r.dw_b[b], r.ocf_w, r.Cp, r.ocf_b, ndp, hs, ws, r.Cp, act_dt, st));
            const float sxf = (float)ins / (4.0f * (float)W), syf = (float)ins / (4.0f * (float)H);
            RUN(refiner_apply_delta_launch(delta, flow, cert, M, sxf, syf, st));
            composed = true;
            final_done = true;
            break;
          }
          if (b == 8 && compose_out_conv && !fused) {  // wide scales: depthwise kernel, then out_conv with the composed weights
            // last block: depthwise + BN + ReLU only; its 1x1 lives inside the composed out_conv (RefinerW::oc_w)
            RUN(dwconv5x5_launch(dcur, dalt, r.dw_w[b], r.dw_b[b], ndp, hs, ws, r.Cp, act_dt, st));
            std::swap(dcur, dalt);
            if (int rc = CK(tp + "_dw" + std::to_string(b), dcur, (size_t)M * r.Cp * esz)) return rc;
            composed = true;
            break;
          }
          if (fused) {  // narrow scales: dw5x5 + 1x1 in one pass over HBM (refiner_block.hip)
            RUN(refiner_block_launch(dcur, dalt, r.dw_w[b], r.dw_b[b], r.pw[b].w, r.pw[b].ldw, r.pw[b].b, ndp, hs, ws,
                                     r.Cp, act_dt, st));
            std::swap(dcur, dalt);
            if (int rc = CK(tp + "_blk" + std::to_string(b), dcur, (size_t)M * r.Cp * esz)) return rc;
            continue;
          }
          if (fuse_refiner_blocks && !dry && refiner_block_wide_supported(r.Cp, act_dt)) {  // C = 576: one kernel, all couts per workgroup
            const int rcw = refiner_block_wide_try_launch(dcur, dalt, r.dw_w[b], r.dw_b[b], r.pw[b].w, r.pw[b].ldw, r.pw[b].b, ndp,
                                                          hs, ws, r.Cp, act_dt, st);
            if (rcw < 0) return rcw;
            if (rcw == 0) {
              std::swap(dcur, dalt);
              if (int rc = CK(tp + "_blk" + std::to_string(b), dcur, (size_t)M * r.Cp * esz)) return rc;
              continue;
            }
          }
          RUN(dwconv5x5_launch(dcur, dalt, r.dw_w[b], r.dw_b[b], ndp, hs, ws, r.Cp, act_dt, st));
          if (int rc = CK(tp + "_dw" + std::to_string(b), dalt, (size_t)M * r.Cp * esz)) return rc;
          GemmArgs g;
          g.A = dalt; g.lda = r.Cp; g.W = r.pw[b].w; g.ldw = r.pw[b].ldw; g.C = dcur; g.ldc = r.Cp;
          g.M = (int)M; g.N = r.Cp; g.K = r.Cp; g.in_dt = act_dt; g.out_dt = act_dt; g.bias = r.pw[b].b;
          g.n_alg = g.k_alg = r.C;  // FLOPs of the reference's C x C convolution, not of the padded one
          RUN(gemm_launch(g, st));
          if (int rc = CK(tp + "_blk" + std::to_string(b), dcur, (size_t)M * r.Cp * esz)) return rc;
        }
        const float sx = (float)ins / (4.0f * (float)W), sy = (float)ins / (4.0f * (float)H);
        if (!final_done)
          RUN(refiner_out_launch(dcur, r.Cp, act_dt, composed ? r.oc_w : r.out_w, composed ? r.oc_b : r.out_b, flow, cert, M, r.Cp,
                                 sx, sy, st));
        if (int rc = CK(tp + "_flow", flow, (size_t)M * 2 * 4)) return rc;
        if (int rc = CK(tp + "_cert", cert, (size_t)M * 4)) return rc;
      }
      if (debug && !dry) {
        const std::string pfx = std::string("p") + (up ? "2" : "1");
        if (int rc = dbg_save((pfx + "_flow" + SCALES[si]).c_str(), flow, (size_t)ndp * hw * 2 * 4, st)) return rc;
        if (int rc = dbg_save((pfx + "_cert" + SCALES[si]).c_str(), cert, (size_t)ndp * hw * 4, st)) return rc;
      }
      if (fw && !dry) {  // corresps[ins] = {"flow", "certainty"} (matcher.py:496-512), before the resize to the next scale
        if (fw->flow[si]) ROMA_CHECK_HIP(hipMemcpyAsync(fw->flow[si], flow, (size_t)ndp * hw * 2 * 4, hipMemcpyDeviceToDevice, st));
        if (fw->cert[si]) ROMA_CHECK_HIP(hipMemcpyAsync(fw->cert[si], cert, (size_t)ndp * hw * 4, hipMemcpyDeviceToDevice, st));
      }
      // (one of our own kernels rather than a runtime copy node on the stream)
      if (ins == 16) {
        const long nkeep = (long)ndp * hw;
        if (nkeep % 4 == 0) {
          RUN(copy2d_launch(cert, nkeep, DT_F32, cert16_keep, nkeep, DT_F32, 1, (int)nkeep, st));
        } else if (!dry) {
          ROMA_CHECK_HIP(hipMemcpyAsync(cert16_keep, cert, (size_t)nkeep * 4, hipMemcpyDeviceToDevice, st));
        }
      }
      arena.release(smark);
      if (ins != 1) {
        const int nh = H / (ins / 2), nw = W / (ins / 2);
        RUN(resize_bilinear_launch(flow, flow_alt, ndp, hs, ws, nh, nw, 2, st));
        RUN(resize_bilinear_launch(cert, cert_alt, ndp, hs, ws, nh, nw, 1, st));
        std::swap(flow, flow_alt);
        std::swap(cert, cert_alt);
        if (int rc = CK(tp + "_flow_up", flow, (size_t)ndp * nh * nw * 2 * 4)) return rc;
        ch = nh; cw = nw;
      }
    }
    (void)ch; (void)cw;
    // the finest flow / certainty of this pass stay where they are
    flow_fin = flow; cert_fin = cert;
    if (!up) { flow_p1 = flow; cert_p1 = cert; }
    arena.release(pass_mark);
  }
  if (!dry && (arena.overflow || persist.overflow)) {
    arena.overflow = persist.overflow = false;
    set_error("roma_match / roma_forward: the workspace planned by roma_finalize is too small for this call (an option that "
              "changes the buffer plan was switched after roma_finalize); the results of this call are invalid");
    return -5;
  }
  if (fw) return 0;
  // =============================== epilogue (matcher.py:839-850, 891-929)
  FinalArgs fa;
  fa.flow = flow_fin;
  fa.cert = cert_fin;
  fa.cert16 = cfg.attenuate_cert ? cert16_keep : nullptr;
  fa.warp = warp_out; fa.certainty = cert_out;
  fa.B = B; fa.H = Hfin; fa.W = Wfin; fa.h16 = th; fa.w16 = tw; fa.symmetric = cfg.symmetric;
  RUN(final_epilogue_launch(fa, st));
  if (int rc = CK("final_warp", warp_out, (size_t)B * Hfin * Wfin * (cfg.symmetric ? 2 : 1) * 4 * 4)) return rc;
  if (int rc = CK("final_cert", cert_out, (size_t)B * Hfin * Wfin * (cfg.symmetric ? 2 : 1) * 4)) return rc;
  return 0;
}

}  // namespace roma
