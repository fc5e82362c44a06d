// Flash-style attention for gfx950 (see attention.h).
//
// One workgroup = 4 wave64 = 128 queries of one (batch, head); each wave owns 32 queries.
// Everything is computed TRANSPOSED so that softmax statistics are lane-local:
//   S^T[key][q] = K . Q^T      (MFMA A = K rows from LDS, B = Q rows held in registers)
//   O^T[d][q]  += V^T . P^T    (MFMA A = V^T rows from LDS, B = P^T straight from the S^T registers)
// The MFMA C/D layout gives lane (q = lane&31, h = lane>>5) the scores of query q against
// 16 of each 32 keys, which is exactly the B-operand layout the second MFMA needs (the key
// order inside a k-step is a free permutation as long as V^T is read with the same one),
// so P never leaves registers and the running max / rescale factor is one scalar per lane.
#include "attention.h"

#include <stdlib.h>
#include <type_traits>
#include <stdio.h>
#include "gemm.h"  // DT_*

namespace roma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// Work item of a workgroup = (query tile qt, head, image b).  Workgroup g runs on XCD g % 8 and every XCD has its own L2,
// so with the plain (qt fastest) order the 13 query tiles of one (b, head) land on 8 different XCDs and EVERY XCD streams
// EVERY head's K / V^T from the fabric (FETCH_SIZE 4.6 x the algorithmic bytes, profiles/r02_pmc_summary.json).  With
// xcd_map the work list is cut into 8 contiguous bands, one per XCD: all query tiles of a (b, head) - and its K / V^T -
// stay on one XCD's L2.
__device__ __forceinline__ bool attn_decode_block(const AttnArgs& a, int& qt, int& head, int& b) {
  const int nq = (a.N + 127) / 128;
  const long n = (long)nq * a.heads * a.B;
  long w = blockIdx.x;
  if (a.xcd_map) {
    const long per = (n + 7) / 8;
    const long slot = blockIdx.x / 8;
    w = (long)(blockIdx.x % 8) * per + slot;
    if (slot >= per || w >= n) return false;
  }
  qt = (int)(w % nq);
  const long bh = w / nq;
  head = (int)(bh % a.heads);
  b = (int)(bh / a.heads);
  return true;
}

template <int HD, typename TOUT>
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnArgs a) {
  constexpr int KV = (HD == 64) ? 64 : 32;
  constexpr int KS = HD + 4, VS = KV + 4;
  constexpr int KT = KV / 32, DT = HD / 32, G = HD / 8;
  __shared__ __attribute__((aligned(16))) float Ks[KV * KS];
  __shared__ __attribute__((aligned(16))) float Vs[HD * VS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  int qt, head, b;
  if (!attn_decode_block(a, qt, head, b)) return;
  const long bh = (long)b * a.heads + head;
  const int qi = qt * 128 + wave * 32 + l31;
  const float* Q = reinterpret_cast<const float*>(a.q) + (bh * a.npad) * HD;
  const float* K = reinterpret_cast<const float*>(a.k) + (bh * a.npad) * HD;
  const float* Vt = reinterpret_cast<const float*>(a.vt) + (bh * HD) * a.npad;

  f32x4 qf[G];
#pragma unroll
  for (int g = 0; g < G; ++g) qf[g] = *reinterpret_cast<const f32x4*>(Q + (long)qi * HD + 8 * g + 4 * h);

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int kv0 = 0; kv0 < a.N; kv0 += KV) {
    // ---- stage K tile [KV][HD] and V^T tile [HD][KV]
    for (int idx = tid; idx < KV * (HD / 4); idx += 256) {
      const int row = idx / (HD / 4), c4 = idx % (HD / 4);
      *reinterpret_cast<f32x4*>(&Ks[row * KS + c4 * 4]) =
          *reinterpret_cast<const f32x4*>(K + (long)(kv0 + row) * HD + c4 * 4);
    }
    for (int idx = tid; idx < HD * (KV / 4); idx += 256) {
      const int row = idx / (KV / 4), c4 = idx % (KV / 4);
      *reinterpret_cast<f32x4*>(&Vs[row * VS + c4 * 4]) =
          *reinterpret_cast<const f32x4*>(Vt + (long)row * a.npad + kv0 + c4 * 4);
    }
    __syncthreads();
    // ---- S^T = K Q^T
    f32x16 s[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(32 * kt + l31) * KS + 8 * g + 4 * h]);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[g][j], s[kt], 0, 0, 0);
      }
    }
    // ---- online softmax (per query = per lane column)
    float tmax = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float sv = key < a.N ? s[kt][r] : -INFINITY;
        s[kt][r] = sv;
        tmax = fmaxf(tmax, sv);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = expf(m_run - m_new);
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf(s[kt][r] - m_new);
        s[kt][r] = p;
        lsum += p;
      }
    l_run = l_run * alpha + lsum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const f32x4 vf = *reinterpret_cast<const f32x4*>(&Vs[(32 * d + l31) * VS + 32 * kt + 8 * rg + 4 * h]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[j], s[kt][4 * rg + j], o[d], 0, 0, 0);
        }
    __syncthreads();
  }
  l_run += __shfl_xor(l_run, 32);
  const float inv = 1.f / l_run;
  if (qi < a.N) {
    TOUT* O = reinterpret_cast<TOUT*>(a.out) + ((long)b * a.N + qi) * a.ldo + head * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = o[d][4 * rg + j] * inv;
        ElemIO<TOUT>::st4(O + 32 * d + 8 * rg + 4 * h, v);
      }
  }
}

// The 16-bit kernel (round 3's "version 2"; the round-1 kernel it replaced - always-rescaling online softmax, K / V staged
// through registers - was removed in round 4 after a round of green history: git history, commit 069022a).
//
// The kernel above is bound by its VALU work, not by the matrix cores: per 64-key tile and wave, 16 MFMAs (512 cycles)
// stand beside ~165 plain VALU instructions + 33 v_exp (profiles/r03_pmc_sq_summary.json: MFMA busy 0.37, the waves wait
// on instruction issue 0.44).  Three of its four O(tile) VALU passes are removed here:
//   * "s - m" (32 v_sub): the running reference m_ref enters the score MFMAs through their C operand - a 16-register
//     block holding -m_ref - so the accumulators come out as q.k - m_ref and go straight into v_exp;
//   * "o *= alpha" (16 v_pk_mul) and the per-tile alpha: m_ref is only moved on the first tile and when some query of the
//     wave has a tile whose P = 2^(s - m_ref) sum to more than 2^14 (rounds 3-5: when a score was more than 2^8 above the
//     reference; the test moved from the scores to the row sums late in round 6, see the tile body).  Until then P is well
//     inside bf16 / f16, and numerator and denominator carry the same factor 2^(-m_ref), which cancels in O / l.  The
//     decision is taken after the previous tile's P.V is complete and before this tile's P enters l or O, so everything at
//     the old reference (O, l) is scaled exactly once and nothing at the new one is;
//   * (the row sums were also tried on the matrix cores - l as one more row block of the P.V product, an A fragment whose
//     row 0 is all ones: 4 more MFMAs and 16 more registers per tile for 32 v_add less.  Slower at both head sizes,
//     5.37 -> 5.48 ms and 1.10 -> 1.14 ms per step (profiles/r03_v9_attention.log): with the three passes above gone the
//     loop is no longer short of VALU slots.  Not kept.)
// The K / V tiles are double buffered in LDS (one barrier per tile instead of two).  Also measured and not kept: all V^T
// fragments of a tile requested before the softmax instead of one right before each MFMA - no change at three workgroups
// per CU (5.239 vs 5.248 ms, profiles/r03_v11_attention.log), and slower when the 32 extra registers cost the third workgroup (v2 / v1 = 0.956 instead of 0.907, profiles/r03_v10_attention.log).
// Neither did a software-pipelined form - the score MFMAs of tile t + 1 issued ahead of the softmax of tile t into a second
// accumulator set, K one tile ahead of V through the buffers: correct (same tests), but its 206 registers leave two
// workgroups per CU and it ran 5.65-5.69 ms against 5.27-5.28 (profiles/r03_v18_attention_pipelined.log): a third wave per
// SIMD hides the softmax better than the wave's own look-ahead.
// K / V addressing is a uniform tile pointer + a per-thread 32-bit offset (the kernel above rebuilt 64-bit addresses
// every tile and spilled 8-10 registers at its 128-register cap).
__device__ __forceinline__ float attn_max_halves(float t) {  // max of lane i and lane i ^ 32, in both
#if defined(__HIP_DEVICE_COMPILE__)
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#else
  return t;
#endif
}

template <int HD, typename TOUT, bool EXP2>
__global__ __launch_bounds__(256, HD == 64 ? 3 : 2) void attn_h16_v2_kernel(const AttnArgs a) {
  constexpr int KV = 64;
  constexpr int KS = HD + 8, VS = KV + 4;  // LDS row pitches, as above
  constexpr int KT = KV / 32, DT = HD / 32, NS = HD / 16;
  // two tile buffers: tile t + 1 is staged while tile t is computed - ONE barrier per tile (36 / 70 KiB per workgroup)
  __shared__ __attribute__((aligned(16))) bf16_t Ksb[2][KV * KS];
  __shared__ __attribute__((aligned(16))) bf16_t Vsb[2][HD * VS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  int qt, head, b;
  if (!attn_decode_block(a, qt, head, b)) return;
  const long bh = (long)b * a.heads + head;
  const int qi = qt * 128 + wave * 32 + l31;
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (bh * a.npad) * HD;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (bh * a.npad) * HD;
  const bf16_t* Vt = reinterpret_cast<const bf16_t*>(a.vt) + (bh * HD) * a.npad;

  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
  u32x4_t qf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const u32x4_t*>(Q + (long)qi * HD + 16 * s + 8 * h);

  f32x16 o[DT], cneg;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d][r] = 0.f;
    cneg[r] = 0.f;
  }
  float m_ref = 0.f, l_run = 0.f;

  // tile = 64 keys: K rows are one contiguous 64 x HD block; V^T is HD rows of 64 consecutive keys
  constexpr int NKC = KV * (HD / 8) / 256, NVC = HD * (KV / 8) / 256;  // 16-byte pieces per thread
  unsigned voff[NVC];  // element offset of this thread's V^T pieces inside the (b, head) slab (< 2^31: HD * npad elements)
#pragma unroll
  for (int i = 0; i < NVC; ++i) {
    const int idx = tid + 256 * i;
    voff[i] = (unsigned)(idx / (KV / 8)) * (unsigned)a.npad + (unsigned)(idx % (KV / 8)) * 8u;
  }
  u32x4_t kreg[NKC], vreg[NVC];
  // Round 6: K / V^T tiles through buffer descriptors over this (image, head)'s slabs - the thread's 32-bit offset is fixed,
  // the tile offset travels in an SGPR: no address arithmetic on the VALU (the loop is bound by its VALU issue slots; the flat
  // form spent five instructions per tile on 64-bit addresses)
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K), 0, (int)((long)a.npad * HD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Vt), 0, (int)((long)a.npad * HD * 2), 0x00020000);
#define ROMA_ATTN_FETCH2(KV0)                                                                      \
  {                                                                                                \
    const int ks_ = (KV0) * (HD * 2), vs_ = (KV0) * 2; /* uniform byte offsets of the tile */       \
    _Pragma("unroll") for (int i = 0; i < NKC; ++i)                                                \
        kreg[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_k, tid * 16 + 4096 * i, ks_, 0)); \
    _Pragma("unroll") for (int i = 0; i < NVC; ++i)                                                \
        vreg[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_v, (int)(voff[i] * 2u), vs_, 0)); \
  }
#define ROMA_ATTN_STAGE(BUF)                                                                        \
  {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < NKC; ++i) {                                              \
      const int idx = tid + 256 * i;                                                               \
      const int row = idx / (HD / 8), c8 = idx % (HD / 8);                                         \
      *reinterpret_cast<u32x4_t*>(&Ksb[BUF][row * KS + c8 * 8]) = kreg[i];                         \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NVC; ++i) {                                              \
      const int idx = tid + 256 * i;                                                               \
      const int row = idx / (KV / 8), c8 = idx % (KV / 8);                                         \
      *reinterpret_cast<u32x2_t*>(&Vsb[BUF][row * VS + c8 * 8]) = u32x2_t{vreg[i].x, vreg[i].y};   \
      *reinterpret_cast<u32x2_t*>(&Vsb[BUF][row * VS + c8 * 8 + 4]) = u32x2_t{vreg[i].z, vreg[i].w}; \
    }                                                                                              \
  }
  ROMA_ATTN_FETCH2(0);
  ROMA_ATTN_STAGE(0);
  ROMA_ATTN_FETCH2(min(KV, a.npad - KV));  // (unconditional fetches: a conditional one made hipcc keep the registers in scratch;
  __syncthreads();                         //  past the end they re-read the last tile)
  // Round 6: the key-tile loop runs in pairs with the LDS buffer index a compile-time constant (a generic lambda instantiated
  // for 0 and 1), so every staging write and fragment read is `base register + immediate` - the loop is bound by its VALU issue
  // slots, and selecting the buffer at run time cost ~10 address instructions per tile.  Same operations in the same order.
  int kv0 = 0;
  // Round 6 (late): the reference-move test leaves the common path.  Tiles after the first exponentiate straight away and look at
  // the row sums they need anyway: a lane's 32 values of P are all <= their sum, so "sum <= 2^14" proves P <= 2^14 (inside
  // binary16's range; bfloat16 has f32's) without the 32-deep max tree, its half-wave exchange and the register copies hipcc
  // put on the not-taken side of the branch (~40 of the ~260 VALU issue slots of a tile).  If some valid query's sum exceeds
  // the limit - or is not finite: 2^x overflows from x = 128 on - the wave re-does the tile the round-3 way: scores again from
  // the K tile still in LDS, tile maximum, O and l moved to the new reference, P at the new reference.  Tile 0 always takes
  // that path's first half (there is no reference yet).  Nothing at the old reference is mixed with anything at the new one:
  // the P of the first attempt is discarded.
  constexpr float LIM = 16384.f;
#define ROMA_ATTN_SCORES()                                                                          \
  _Pragma("unroll") for (int st = 0; st < NS; ++st) _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) { \
    const u32x4_t kf = *reinterpret_cast<const u32x4_t*>(&Ks[(32 * kt + l31) * KS + 16 * st + 8 * h]); \
    s[kt] = mfma_h16_32x32x16(kf, qf[st], st == 0 ? cneg : s[kt]); /* scores relative to m_ref */    \
  }                                                                                                 \
  if (kv0 + KV > a.N) { /* only the last tile has keys >= N to mask (uniform branch) */            \
    /* key = kv0 + 32 kt + 4 h + c_r with c_r = (r & 3) + 8 (r >> 2) a constant: one per-lane limit, constants compared */ \
    /* against it (as `key >= N` per element the index arithmetic was hoisted out of the branch: 35 VALU on every tile)  */ \
    int lim_ = a.N - kv0 - 4 * h;                                                                    \
    asm volatile("" : "+v"(lim_));                                                                  \
    _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) _Pragma("unroll") for (int r = 0; r < 16; ++r)  \
        if (32 * kt + (r & 3) + 8 * (r >> 2) >= lim_) s[kt][r] = -INFINITY;                          \
  }
#define ROMA_ATTN_EXP_SUM()                                                                         \
  _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) _Pragma("unroll") for (int r = 0; r < 16; ++r)    \
      s[kt][r] = EXP2 ? __builtin_amdgcn_exp2f(s[kt][r]) : __expf(s[kt][r]);                        \
  { /* row sums: four independent chains (one 32-deep chain of dependent adds is ~1.7 x slower to issue) */ \
    float ls[4] = {0.f, 0.f, 0.f, 0.f};                                                             \
    _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) _Pragma("unroll") for (int r = 0; r < 16; ++r) ls[r & 3] += s[kt][r]; \
    lsum = (ls[0] + ls[1]) + (ls[2] + ls[3]);                                                       \
  }
  auto tile = [&](auto curc, auto firstc) __attribute__((always_inline)) {
    constexpr int cur = decltype(curc)::value;
    constexpr bool FIRST = decltype(firstc)::value != 0;
    // stage tile t + 1 into the other buffer: every wave finished reading it (tile t - 1) before the barrier that ended
    // the previous iteration; its loads were issued a whole tile ago
    ROMA_ATTN_STAGE(cur ^ 1);
    ROMA_ATTN_FETCH2(min(kv0 + 2 * KV, a.npad - KV));
    const bf16_t* const Ks = Ksb[cur];
    const bf16_t* const Vs = Vsb[cur];
    f32x16 s[KT];
    float lsum = 0.f;
    // the two 32-key chains interleaved: consecutive MFMAs never depend on each other
    ROMA_ATTN_SCORES()
    bool redo = FIRST;
    if constexpr (!FIRST) {
      ROMA_ATTN_EXP_SUM()
      // wave-uniform decision.  Padding queries (qi >= N) have no vote: their Q rows are whatever the workspace holds (the
      // model's DINOv2 and decoder layouts share it, so "padding" of one is data of the other), and a vote of theirs would make
      // the rounding of the VALID queries of the wave depend on that leftover - seen as run-to-run differences of the last
      // patch token, amplified by the coarse arg-max (profiles/r03_v24_*.log).
      redo = __builtin_expect(__builtin_amdgcn_ballot_w64(qi < a.N && !(lsum <= LIM)) != 0, 0);
      if (redo) {
        ROMA_ATTN_SCORES()  // (two statements: scores and mask)
      }
    }
    if (__builtin_expect(redo, FIRST)) {  // always on tile 0, rare afterwards
      float tm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // four independent v_max3 chains, not one of 17
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) tm[(2 * kt + (r >> 3)) & 3] = fmaxf(tm[(2 * kt + (r >> 3)) & 3], s[kt][r]);
      const float tmax = attn_max_halves(fmaxf(fmaxf(tm[0], tm[1]), fmaxf(tm[2], tm[3])));
      const float delta = FIRST ? tmax : fmaxf(tmax, 0.f);
      if constexpr (!FIRST) {  // O and l are still zero on the first tile (and alpha could overflow there)
        const float alpha = EXP2 ? __builtin_amdgcn_exp2f(-delta) : __expf(-delta);
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        l_run *= alpha;
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] -= delta;
      m_ref += delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) cneg[r] = -m_ref;
      ROMA_ATTN_EXP_SUM()
    }
    l_run += lsum;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4_t pk;
        pk.x = pack_bf16x2(s[kt][8 * u + 0], s[kt][8 * u + 1]);
        pk.y = pack_bf16x2(s[kt][8 * u + 2], s[kt][8 * u + 3]);
        pk.z = pack_bf16x2(s[kt][8 * u + 4], s[kt][8 * u + 5]);
        pk.w = pack_bf16x2(s[kt][8 * u + 6], s[kt][8 * u + 7]);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const bf16_t* vrow = &Vs[(32 * d + l31) * VS + 32 * kt + 16 * u + 4 * h];
          const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(vrow);
          const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(vrow + 8);
          o[d] = mfma_h16_32x32x16(u32x4_t{lo.x, lo.y, hi.x, hi.y}, pk, o[d]);
        }
      }
    __syncthreads();
  };
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  tile(c0{}, c1{});  // tile 0 sets the reference
  kv0 += KV;
  while (kv0 < a.N) {
    tile(c1{}, c0{});
    kv0 += KV;
    if (kv0 >= a.N) break;
    tile(c0{}, c0{});
    kv0 += KV;
  }
#undef ROMA_ATTN_SCORES
#undef ROMA_ATTN_EXP_SUM
  float l = l_run;
  l += __shfl_xor(l, 32);
  const float inv = 1.f / l;
  if (qi < a.N) {
    TOUT* O = reinterpret_cast<TOUT*>(a.out) + ((long)b * a.N + qi) * a.ldo + head * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = o[d][4 * rg + j] * inv;
        ElemIO<TOUT>::st4(O + 32 * d + 8 * rg + 4 * h, v);
      }
  }
}
#undef ROMA_ATTN_STAGE

int g_attn_exp2 = -1;     // roma_tuning("attn_exp2", v): tools only - force the 2^x softmax (q pre-scaled by log2 e) on / off; -1 = as the caller says
int g_attn_xcd_map = -1;  // roma_tuning("attn_xcd", v): 1 = per-XCD bands of (b, head) (default), 0 = plain order, -1 = env ROMA_ATTN_XCD

int attention_launch(const AttnArgs& a_in, hipStream_t stream) {
  AttnArgs a = a_in;
  ROMA_REQUIRE(a.hd == 64 || a.hd == 128, "attention: head dim must be 64 or 128");
  ROMA_REQUIRE(a.npad % 128 == 0 && a.npad >= a.N, "attention: Npad must be a multiple of 128 and >= N");
  ROMA_REQUIRE(a.ldo % 4 == 0, "attention: ldo must be a multiple of 4");
  if (g_attn_exp2 >= 0) a.exp2_domain = g_attn_exp2 ? 1 : 0;  // tools/attn_determinism.py (never set on the product path)
  const long nwork = (long)((a.N + 127) / 128) * a.heads * a.B;
  static const int map_env = getenv("ROMA_ATTN_XCD") ? atoi(getenv("ROMA_ATTN_XCD")) : 1;
  a.xcd_map = g_attn_xcd_map >= 0 ? g_attn_xcd_map : map_env;
  ROMA_REQUIRE(nwork > 0 && nwork < (1l << 30), "attention: bad problem size");
  dim3 grid((unsigned)(a.xcd_map ? 8 * ((nwork + 7) / 8) : nwork));
  char pname[64];
  snprintf(pname, sizeof pname, "attn_%s_kernel<%d>", a.in_dt == DT_F32 ? "f32" : ROMA_H16_NAME, a.hd);
  ProfScope ps(pname, 4.0 * (double)a.B * a.heads * (double)a.N * a.N * a.hd, "flop", stream);
#define ROMA_ATTN(KERNEL, HDV, TOUT) hipLaunchKernelGGL((KERNEL<HDV, TOUT>), grid, dim3(256), 0, stream, a)
#define ROMA_ATTNV2(HDV, TOUT, E2) hipLaunchKernelGGL((attn_h16_v2_kernel<HDV, TOUT, E2>), grid, dim3(256), 0, stream, a);
#define ROMA_ATTNB(HDV, TOUT) \
  { if (a.exp2_domain) ROMA_ATTNV2(HDV, TOUT, true) else ROMA_ATTNV2(HDV, TOUT, false) }
  if (a.in_dt == DT_F32) {
    if (a.hd == 64) { if (a.out_dt == DT_F32) ROMA_ATTN(attn_f32_kernel, 64, float); else ROMA_ATTN(attn_f32_kernel, 64, bf16_t); }
    else            { if (a.out_dt == DT_F32) ROMA_ATTN(attn_f32_kernel, 128, float); else ROMA_ATTN(attn_f32_kernel, 128, bf16_t); }
  } else {
    if (a.hd == 64) { if (a.out_dt == DT_F32) ROMA_ATTNB(64, float) else ROMA_ATTNB(64, bf16_t) }
    else            { if (a.out_dt == DT_F32) ROMA_ATTNB(128, float) else ROMA_ATTNB(128, bf16_t) }
  }
#undef ROMA_ATTN
#undef ROMA_ATTNB
#undef ROMA_ATTNV2
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma
