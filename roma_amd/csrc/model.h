// Host-side model object behind the C ABI: state-dict intake, BN folding / repacking, device
// workspace (bump arena, planned by a dry run - no allocation inside roma_match) and the
// kernel schedule of RegressionMatcher.match() (romatch/models/matcher.py:779-934).
#pragma once
#include <stdlib.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/roma_hip.h"
#include "common.h"

namespace roma {

extern std::mutex g_peer_mutex;  // guards the one-time load of the sibling library and the mixed-handle count (model.hip)

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  long numel() const {
    long n = 1;
    for (auto d : shape) n *= d;
    return n;
  }
};

struct Lin {  // y = x W^T + b ; W in activation dtype [N][ldw]
  void* w = nullptr;
  float* b = nullptr;
  int N = 0, K = 0, ldw = 0;
};

struct VitBlockW {
  float *ln1w = nullptr, *ln1b = nullptr, *ln2w = nullptr, *ln2b = nullptr, *ls1 = nullptr, *ls2 = nullptr;
  Lin qkv, proj, fc1, fc2;
  roma_vit_block_t pub() const {  // the plain-pointer form vit.h / roma_vit_forward take
    roma_vit_block_t b{};
    b.ln1_w = ln1w; b.ln1_b = ln1b; b.ln2_w = ln2w; b.ln2_b = ln2b; b.ls1 = ls1; b.ls2 = ls2;
    b.qkv_w = qkv.w; b.proj_w = proj.w; b.fc1_w = fc1.w; b.fc2_w = fc2.w;
    b.qkv_b = qkv.b; b.proj_b = proj.b; b.fc1_b = fc1.b; b.fc2_b = fc2.b;
    b.qkv_ldw = qkv.ldw; b.proj_ldw = proj.ldw; b.fc1_ldw = fc1.ldw; b.fc2_ldw = fc2.ldw;
    return b;
  }
};

struct RefinerW {
  int Cf = 0, E = 0, radius = 0, K = 0, C = 0, Cp = 0;
  float *emb_w = nullptr, *emb_b = nullptr;
  float* dw_w[9] = {nullptr};  // [25][Cp], BN folded
  float* dw_b[9] = {nullptr};
  Lin pw[9];
  float *out_w = nullptr, *out_b = nullptr;  // [3][Cp], [3]
  // out_conv composed with the LAST block's 1x1 (two linear maps with nothing in between, matcher.py:92-122, 175-178):
  // oc_w = out_w . pw[8] ([3][Cp], f32), oc_b = out_w . pw[8].b + out_b - the last block then needs no C x C GEMM at all
  float *oc_w = nullptr, *oc_b = nullptr;
  // the same for the FINAL form of the fused narrow blocks (refiner_block.h): 16-bit [8][Cp], rows 0-2 head / 4-6 remainder of
  // oc_w, and the composed bias zero-padded to [Cp]
  void* ocf_w = nullptr;
  float* ocf_b = nullptr;
};

class Arena {
 public:
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool dry = true;
  // set when a live allocation did not fit the planned capacity (an option changed between roma_finalize's dry run and the
  // call): the request is then served from the START of the arena - aliased, so the call's results are garbage, but no
  // kernel writes outside the hipMalloc'd workspace - and the entry point reports the error instead of returning results
  bool overflow = false;
  void reset() { off = 0; }
  void* alloc(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    if (!dry && off + bytes > cap) {
      overflow = true;
      if (off > peak) peak = off;
      return bytes <= cap ? (void*)base : nullptr;
    }
    void* p = dry ? reinterpret_cast<void*>((uintptr_t)0x1000 + off) : (void*)(base + off);
    off += bytes;
    if (off > peak) peak = off;
    return p;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

class Model {
 public:
  static constexpr int MAX_STREAMS_DECL = 4;
  roma_config_t cfg{};
  int act_dt = 0;  // DT_F32 / DT_BF16
  // ROMA_MIXED (binary16 build only): DINOv2 runs in bfloat16 in the sibling library (its weights are packed as bfloat16
  // bits here, its patch tokens converted to binary16 behind it), everything else in this build's binary16.
  bool mixed = false;
  bool pack_as_bf16 = false;                                   // upload_act: write bfloat16 bits whatever the build
  void* peer_lib = nullptr;                                    // dlopen handle of libroma_hip.so
  int (*peer_vit_forward)(const roma_vit_args_t*, void*) = nullptr;
  bool finalized = false;
  bool debug = false;
  // The last ConvRefiner block's 1x1 convolution and out_conv are composed into ONE C -> 3 map at pack time (option
  // "compose_out_conv", default 1): 1/9 of the refiners' 1x1 GEMM work and one pass over the block output disappear; the
  // result differs from the two-step evaluation only by rounding (the 16-bit modes no longer round the dropped intermediate)
  bool compose_out_conv = !(getenv("ROMA_COMPOSE_OUT") && atoi(getenv("ROMA_COMPOSE_OUT")) == 0);  // env: A/B runs
  // VGG layers with Cout >= 256 in the 16-bit modes: weight rows in slab-major K order (gemm.h, GemmArgs::conv_korder) - the nine
  // taps of a 64-channel slab back to back, an L2-sized working set per workgroup.  Measured in round 5: L2-miss traffic of the
  // conv GEMM 477 -> 143 MB raw FETCH_SIZE per launch, and the kernel 1.6 % SLOWER (9.31 against 9.15 ms per step, three
  // alternations on one box: profiles/r05_v15_conv_k_order_ab.log) - the misses were Infinity-Cache hits and the tap-major walk
  // reads each pixel's channels as one contiguous run.  OFF by default; ROMA_CONV_KORDER=1 selects it (read before the weights
  // are packed; it changes only the summation order of the K loop).
  // Round 6: ON by default - it is the K order of the patch-resident kernel (conv_patch.hip), which keeps the activation patch of a
  // slab in LDS across the nine taps.  ROMA_CONV_KORDER=0 (or ROMA_CONV_PATCH=0) restores tap-major rows on the implicit GEMM.
  bool vgg_slab_major = !(getenv("ROMA_CONV_KORDER") && atoi(getenv("ROMA_CONV_KORDER")) == 0) &&
                        !(getenv("ROMA_CONV_PATCH") && atoi(getenv("ROMA_CONV_PATCH")) == 0);
  int vgg_korder[12] = {0};
  bool fuse_refiner_blocks = true;  // bf16 mode: fused dw5x5+1x1 kernel at the narrow scales (option "fuse_refiner_blocks")
  // bf16 mode: DINOv2's residual stream in bf16, like the reference's bf16 backbone (encoders.py: dinov2 weights and
  // input are cast to amp_dtype); the decoder transformer keeps f32 (autocast leaves its residual f32).  Option
  // "vit_bf16_residual"; env ROMA_VIT_RES_F32=1 forces the f32 stream for A/B runs.
  bool vit_bf16_residual = true;
  // option "trace": every stage of match() XORs a checksum of its output into a per-stream table (roma_debug_trace)
  bool trace_on = false;
  static constexpr int TRACE_MAX = 1024;
  unsigned long long* trace_dev[MAX_STREAMS_DECL] = {nullptr};
  std::vector<std::string> trace_names[MAX_STREAMS_DECL];
  int trace_n[MAX_STREAMS_DECL] = {0};
  double coarse_scale_factor = 0.0;  // roma_set_option_f; 0 = sqrt(coarse_h * coarse_w / 560^2)
  std::map<std::string, HostTensor> host;

  // packed weights (device)
  float *c1_w = nullptr, *c1_b = nullptr;  // first VGG conv [27][64]
  Lin vgg[12];                             // [1..11] implicit-GEMM convs (index 0 unused)
  int vgg_cin[12] = {0}, vgg_cout[12] = {0};
  Lin patch;
  float *cls_tok = nullptr, *pos_emb = nullptr;  // pos-embed already resized to the coarse token grid
  VitBlockW dino[24];
  float *dino_nw = nullptr, *dino_nb = nullptr;
  VitBlockW tdec[5];
  Lin to_out;
  float *gp_w = nullptr, *gp_b = nullptr;
  Lin proj[5];  // scales 16,8,4,2,1
  RefinerW ref[5];

  Arena arena;        // per-call scratch
  Arena persist;      // zero-initialised, never aliased (attention q/k/v^T pads, GP basis)
  // Sub-batches on several HIP streams (option "streams" 1..4, "dual_stream" = 2 / 1; env ROMA_STREAMS overrides):
  // pairs are independent, so sub-batch i > 0 runs the same schedule out of its own arenas on side stream i and the
  // partially filled last rounds of one sub-batch's kernels (e.g. 404 GEMM tiles on 256 CUs) are filled by the
  // others' work: +5 % at batch 8 with 2 streams (profiles/r02_final_bench_bf16.json vs r02_final_bench_bf16_1stream.json).
  // ON by default (2 streams) since round 2: results are bit-identical to the single-stream schedule (2 000 / 2 000 bf16
  // stress runs in the GPU suite; the 1-5 % one-ulp deviations of round 1 went away with the GEMM epilogue rewrite,
  // while the library of the commit before it still shows them on the same box - DESIGN.md section 4).
  // Side arenas / streams are created on first use from the sizes planned at roma_finalize.
  static constexpr int MAX_STREAMS = 4;
  int n_streams = 2, streams_ready = 1;
  size_t side_arena_bytes = 0, side_persist_bytes = 0;
  Arena side_arena[MAX_STREAMS - 1], side_persist[MAX_STREAMS - 1];
  hipStream_t side[MAX_STREAMS - 1] = {nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[MAX_STREAMS - 1] = {nullptr};
  std::vector<void*> owned;  // device allocations to free
  std::map<std::string, std::pair<void*, size_t>> dbg;
  int dbg_cur_slot = 0;  // sub-batch slot match_impl is currently enqueueing (ROMA_DEBUG_DUAL_SLOT diagnostics)
  // debug mode only (roma_debug_inject): device buffers that REPLACE a named intermediate of the next match() calls.
  // "gm_flow16" [ndp, T, 2] / "gm_cert16" [ndp, T] f32 overwrite the output of cls_to_flow_refine, so a parity test can
  // pin the (discontinuous) coarse arg-max to the oracle's and hold everything downstream to a continuous bound.
  std::map<std::string, std::pair<void*, size_t>> inject;
  int debug_inject(const char* name, const void* host, size_t bytes);

  ~Model();
  int set_tensor(const char* name, int ndim, const int64_t* shape, const void* data, int is_int64);
  int finalize();
  int match(int B, const float* ima, const float* imb, const float* ima_hr, const float* imb_hr, float* warp,
            float* cert, hipStream_t st);
  int forward(int B, const float* ima, const float* imb, const roma_forward_args_t* fw, hipStream_t st);

 private:
  int ensure_side_streams(int n);
  int match_streams(int B, const float* ima, const float* imb, const float* ima_hr, const float* imb_hr, float* warp,
                    float* cert, hipStream_t st);
  int check_contract();
  int pack_weights();
  int load_peer();
  // fw != nullptr: ONE pass (roma_forward) - per-scale outputs copied out, no epilogue
  int match_impl(int B, const float* ima, const float* imb, const float* ima_hr, const float* imb_hr, float* warp,
                 float* cert, hipStream_t st, bool dry, Arena& arena, Arena& persist, const roma_forward_args_t* fw = nullptr);
  int dbg_save(const char* name, const void* p, size_t bytes, hipStream_t st);
  template <typename F> int upload_f32(const std::vector<float>& v, F** out);
  int upload_act(const std::vector<float>& v, void** out);
  int make_lin(const std::vector<float>& w, const std::vector<float>* b, int N, int K, Lin* out);
};

// GP match encoder for all directed pairs of a call (model.hip); scratch comes from `arena` (dry = plan sizes only)
int gp_posterior(const void* pf, long ldf, int act_dt, int B, bool symmetric, int th, int tw, const float* gp_w,
                 const float* gp_b, float* mu, long ld_mu, Arena& arena, hipStream_t st, bool dry);

// Batched SPD solve via blocked Cholesky (see include/roma_hip.h roma_op_cholesky_solve_t)
// strideA / strideR: batch strides of A and Rt in floats (0 = dense: n * n and d * n).  When Rt is stored right behind A
// (Rt == A + n * n, strideA == strideR: one (n + d) x n matrix per item) the forward substitution is folded into the
// factorisation loop - see cholesky_solve_t.
int cholesky_solve_t(float* A, float* Rt, float* LT, float* Linv, float* LinvT, int n, int d, int batch, hipStream_t st,
                     long strideA = 0, long strideR = 0);

}  // namespace roma
